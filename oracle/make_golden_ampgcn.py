"""Generate tests/golden/model_*.npz from the REFERENCE's AMPGCN.

TEST INFRASTRUCTURE; runs only in the build container (it needs /root/reference).  The reference
files src/ampnet/module/amp_gcn.py and src/ampnet/conv/amp_conv.py are loaded UNMODIFIED by file
path.  Stand-ins exist only for third-party / plotting imports that are not installed here and
that the forward/backward path never calls with an effect:

  torch_geometric.nn.MessagePassing   aggr='mean' base class (as in oracle/make_golden.py; semantics
                                      pinned by synthetic_benchmark/testing_message_passing_pyg.py:37-40)
  torch_geometric.datasets.Planetoid  imported at amp_gcn.py:14, never used by the class
  torch_geometric.utils.dropout.dropout_adj   amp_gcn.py:15,241: with p = 0 (the Cora harness,
                                      cora_benchmark_graphsaint.py:70-71) it returns the edges unchanged
  seaborn                             amp_gcn.py:6, plots only
  src.ampnet.utils.utils              amp_gcn.py:17 `import *`: plotting helpers + accuracy(), unused here

Fixtures hold data only: node features, edge_index, the state dict, the feature indices the
reference's np.random.choice drew (`sampled_node_feat_indices`), its logits, its two layer
embeddings and every parameter gradient of sum(logits * dlogits).  Two configurations:
  cora  experiments/cora_benchmark_graphsaint.py:59-73  (D=128, H=4, L=20, 1433 features, down-sampling,
        average pooling, dropout 0)
  xor   synthetic_benchmark/xor_training_utils.py:58-72 (D=3, H=1, L=2, downsample_feature_vectors=False,
        feature_repeats=1) and the same with average_pooling_flag=False (token-0 pooling, amp_gcn.py:268-271)

    python oracle/make_golden_ampgcn.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = '/root/reference/src/ampnet'
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def load_reference_ampgcn():
    class MessagePassing(nn.Module):          # stand-in for PyG's base class only
        def __init__(self, aggr='mean'):
            super().__init__()
            self.aggr = aggr

        def propagate(self, edge_index, x):
            src, dst = edge_index[0], edge_index[1]
            m = self.message(x_i=x.index_select(0, dst), x_j=x.index_select(0, src))
            out = torch.zeros(x.size(0), m.size(1), dtype=m.dtype).index_add_(0, dst, m)
            cnt = torch.zeros(x.size(0), dtype=m.dtype).index_add_(0, dst, torch.ones_like(dst, dtype=m.dtype))
            return out / cnt.clamp(min=1).unsqueeze(-1)

    def dropout_adj(edge_index, p=0.0, training=True, **kw):
        assert p == 0.0 or not training, 'the fixtures are generated without adjacency dropout'
        return edge_index, None

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tg = module('torch_geometric')
    tg.nn = module('torch_geometric.nn', MessagePassing=MessagePassing)
    tg.datasets = module('torch_geometric.datasets', Planetoid=None)
    tg.utils = module('torch_geometric.utils')
    tg.utils.dropout = module('torch_geometric.utils.dropout', dropout_adj=dropout_adj)
    module('seaborn')
    for pkg in ('src', 'src.ampnet', 'src.ampnet.conv', 'src.ampnet.utils', 'src.ampnet.module'):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    module('src.ampnet.utils.utils')
    for name, path in (('src.ampnet.conv.amp_conv', f'{REF}/conv/amp_conv.py'),
                       ('src.ampnet.module.amp_gcn', f'{REF}/module/amp_gcn.py')):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return sys.modules['src.ampnet.module.amp_gcn'].AMPGCN


def run(AMPGCN, name, x, edge_index, seed, **cfg):
    torch.manual_seed(seed)
    np.random.seed(seed)                        # the reference samples with np.random.choice (amp_gcn.py:135)
    model = AMPGCN(device='cpu', **cfg)
    with torch.no_grad():                       # exercise the bias paths (default init is 0)
        for conv in (model.conv1, model.conv2):
            conv.multi_head_attention.in_proj_bias.normal_(0, 0.1)
            conv.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    model.train()                               # dropout rates are 0: train == eval arithmetic
    data = types.SimpleNamespace(x=torch.from_numpy(x), edge_index=torch.from_numpy(edge_index))
    logits = model(data)
    g = torch.Generator().manual_seed(seed + 1)
    dlogits = torch.randn(logits.shape, generator=g)
    (logits * dlogits).sum().backward()
    out = dict(x=x, edge_index=edge_index, logits=logits.detach().numpy(), dlogits=dlogits.numpy(),
               conv1_embedding=model.conv1_embedding.detach().numpy(),
               conv2_embedding=model.conv2_embedding.detach().numpy(),
               cfg_keys=np.array(sorted(cfg)), cfg_vals=np.array([str(cfg[k]) for k in sorted(cfg)]))
    if model.sampled_node_feat_indices is not None:
        out['sampled_node_feat_indices'] = np.asarray(model.sampled_node_feat_indices, dtype=np.int64)
    for k, v in model.state_dict().items():
        out['param.' + k] = v.detach().numpy().copy()
    for k, p in model.named_parameters():
        if p.grad is not None:
            out['grad.' + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print(name, 'logits', tuple(logits.shape), 'params', sum(p.numel() for p in model.parameters()))


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    AMPGCN = load_reference_ampgcn()
    rng = np.random.default_rng(20221006)

    # Cora harness configuration on a Cora-shaped bag-of-words sample (Cora itself is a download)
    N, Fdim = 48, 1433
    x = (rng.random((N, Fdim)) < 0.013).astype(np.float32)
    x[np.arange(N), rng.integers(0, Fdim, N)] = 1.0                    # every node has a present word
    src = rng.integers(0, N, 150)
    dst = rng.integers(0, N - 4, 150)                                   # the last 4 nodes receive nothing
    ei = np.concatenate([np.stack([src, dst]), np.stack([dst[:40], src[:40]])], axis=1).astype(np.int64)
    run(AMPGCN, 'model_cora', x, ei, seed=31, embedding_dim=128, num_heads=4, num_node_features=1433,
        num_sampled_vectors=20, output_dim=7, softmax_out=True, feat_emb_dim=127, val_emb_dim=1,
        downsample_feature_vectors=True, average_pooling_flag=True, dropout_rate=0.0, dropout_adj_rate=0.0,
        feature_repeats=None)

    # XOR configuration (xor_training_utils.py:58-72 with feature_repeats = 1: two features per node)
    Nx = 64
    bits = rng.integers(0, 2, (Nx, 2))
    xx = (bits + rng.normal(0, 0.1, (Nx, 2))).astype(np.float32)
    ex = np.stack([rng.integers(0, Nx, 256), rng.integers(0, Nx, 256)]).astype(np.int64)
    xor_cfg = dict(embedding_dim=3, num_heads=1, num_node_features=2, num_sampled_vectors=2, output_dim=2,
                   softmax_out=True, feat_emb_dim=2, val_emb_dim=1, downsample_feature_vectors=False,
                   dropout_rate=0.0, dropout_adj_rate=0.0, feature_repeats=1)
    run(AMPGCN, 'model_xor', xx, ex, seed=32, average_pooling_flag=True, **xor_cfg)
    run(AMPGCN, 'model_xor_tok0', xx, ex, seed=33, average_pooling_flag=False, **xor_cfg)


if __name__ == '__main__':
    main()
