"""One-subgraph-per-GPU data parallelism: gradient averaging over RCCL.

Mirrors what experiments/cora_benchmark_graphsaint_distributed.py:63-66,83-94
INTENDS (DDP over per-rank GraphSAINT subgraphs; the script calls the un-wrapped
model at :83 so its reducer never arms -- SURVEY.md section 3.3).  Here every rank
draws its own subgraph, runs forward/backward locally, and one all-reduce of ONE
flat fp32 buffer (<= ~2 MB for two AMPConv layers at D=256: latency-bound, so a
single collective, no bucketing) averages the gradients before optimizer.step().

Backend "nccl" is RCCL on ROCm (xGMI between the 8 MI355X of a node); "gloo" is
used by the CPU tests.
"""
import torch
import torch.distributed as dist


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (what
    DistributedDataParallel does at construction, ..._distributed.py:63)."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class GradientAllReducer:
    """Flat-buffer mean all-reduce of `params`' gradients.

        reducer = GradientAllReducer(model.parameters())
        loss.backward(); reducer.allreduce(); optimizer.step()
    """

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self._flat = None

    def _buffer(self, like):
        if self._flat is None or self._flat.device != like.device:
            self._flat = torch.empty(self.numel, dtype=torch.float32, device=like.device)   # fp32 sum
        return self._flat

    def allreduce(self, average=True):
        """average=True: mean over ranks (independent mini-batches per rank); False: sum (ranks hold
        partitions of ONE graph, ampnet_amd/partitioned.py)."""
        if not self.params:
            return None
        world = dist.get_world_size(self.group)
        flat = self._buffer(self.params[0])
        off = 0
        for p in self.params:                       # pack (missing grads count as zero)
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)      # also as the only rank: the same code path
        if average and world > 1:
            flat.mul_(1.0 / world)
        off = 0
        for p in self.params:                       # unpack
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.to(p.dtype).clone()
            else:
                p.grad.copy_(g)
            off += n
        return flat
