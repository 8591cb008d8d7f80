"""ampnet_amd -- MI355X-native AMPConv (the hot path of HarryL-Git/ampnet).

    from ampnet_amd import AMPConv          # drop-in for src.ampnet.conv.AMPConv

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).
Compute: libampconv.so, hand-written HIP for gfx950 behind the C ABI of
include/ampconv.h.  No CPU fallback.
"""
from .conv import AMPConv, InvalidConfiguration
from .graph import EdgeCSR, graph_cache
from . import distributed
from .module import AMPGCN, FeatureTokens
from .sampler import GraphSAINTRandomWalkSampler
from .partitioned import NodePartition, PartitionedAMPConv
from .graphed import GraphedAMPConv

__all__ = ['AMPConv', 'InvalidConfiguration', 'EdgeCSR', 'graph_cache', 'distributed', 'AMPGCN', 'FeatureTokens',
           'GraphSAINTRandomWalkSampler', 'NodePartition', 'PartitionedAMPConv', 'GraphedAMPConv']
