"""Generate tests/golden/*.npz from the REFERENCE implementation.

TEST INFRASTRUCTURE; runs only in the build container (it needs
/root/reference).  The reference file src/ampnet/conv/amp_conv.py is loaded
unmodified by file path; the only stand-in is for the un-installed third-party
base class torch_geometric.nn.MessagePassing (aggr='mean'), whose semantics the
reference pins in synthetic_benchmark/testing_message_passing_pyg.py:37-40.
Nothing of the reference is copied into the fixtures: they hold inputs
(x, edge_index, parameters, upstream gradient) and the outputs the reference
produced for them.

    python oracle/make_golden.py                  # rewrites tests/golden/
    python oracle/make_golden.py --linear-only    # only the softmax-free fixtures (linear_*.npz)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF_FILE = '/root/reference/src/ampnet/conv/amp_conv.py'
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def load_reference():
    class MessagePassing(nn.Module):          # stand-in for PyG's base class only
        def __init__(self, aggr='mean'):
            super().__init__()
            self.aggr = aggr

        def propagate(self, edge_index, x):
            src, dst = edge_index[0], edge_index[1]
            m = self.message(x_i=x.index_select(0, dst), x_j=x.index_select(0, src))
            out = torch.zeros(x.size(0), m.size(1), dtype=m.dtype).index_add_(0, dst, m)
            cnt = torch.zeros(x.size(0), dtype=m.dtype).index_add_(
                0, dst, torch.ones_like(dst, dtype=m.dtype))
            return out / cnt.clamp(min=1).unsqueeze(-1)

    tg, tgnn = types.ModuleType('torch_geometric'), types.ModuleType('torch_geometric.nn')
    tgnn.MessagePassing = MessagePassing
    tg.nn = tgnn
    sys.modules['torch_geometric'], sys.modules['torch_geometric.nn'] = tg, tgnn
    spec = importlib.util.spec_from_file_location('ref_amp_conv', REF_FILE)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


CUSTOM_MHA = '/root/reference/src/ampnet/conv/custom_multihead_attn.py'
CUSTOM_FWD = '/root/reference/src/ampnet/conv/custom_multihead_attn_forward.py'


def load_softmax_free_mha():
    """The reference's own softmax-free nn.MultiheadAttention copy (custom_multihead_attn.py,
    custom_multihead_attn_forward.py:4179-4180), loaded unmodified by file path under the module
    names it imports itself by (its package __init__ pulls in un-installed plotting libraries)."""
    for pkg in ('src', 'src.ampnet', 'src.ampnet.conv'):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    mods = {}
    for name, path in (('src.ampnet.conv.custom_multihead_attn_forward', CUSTOM_FWD),
                       ('src.ampnet.conv.custom_multihead_attn', CUSTOM_MHA)):
        spec = importlib.util.spec_from_file_location(name, path)
        mods[name] = importlib.util.module_from_spec(spec)
        sys.modules[name] = mods[name]
        spec.loader.exec_module(mods[name])
    return mods['src.ampnet.conv.custom_multihead_attn'].MultiheadAttention


def make_layer(ref, D, H, seed, mha_cls=None):
    torch.manual_seed(seed)
    layer = ref.AMPConv(embed_dim=D, num_heads=H)
    if mha_cls is not None:
        # what un-commenting amp_conv.py:6 and using it at :18 does: the layer's attention module
        # becomes the reference's softmax-free class (same parameters, same constructor arguments)
        torch.manual_seed(seed)
        layer.multi_head_attention = mha_cls(embed_dim=D, num_heads=H, batch_first=True, bias=True)
    with torch.no_grad():                      # exercise the bias paths (default init is 0)
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    return layer


def params_of(layer, prefix=''):
    m = layer.multi_head_attention
    return {prefix + 'in_proj_weight': m.in_proj_weight.detach().numpy().copy(),
            prefix + 'in_proj_bias': m.in_proj_bias.detach().numpy().copy(),
            prefix + 'out_proj_weight': m.out_proj.weight.detach().numpy().copy(),
            prefix + 'out_proj_bias': m.out_proj.bias.detach().numpy().copy()}


def grads_of(layer, prefix=''):
    m = layer.multi_head_attention
    return {prefix + 'g_in_proj_weight': m.in_proj_weight.grad.numpy().copy(),
            prefix + 'g_in_proj_bias': m.in_proj_bias.grad.numpy().copy(),
            prefix + 'g_out_proj_weight': m.out_proj.weight.grad.numpy().copy(),
            prefix + 'g_out_proj_bias': m.out_proj.bias.grad.numpy().copy()}


def random_graph(rng, N, E, zero_in=0, dups=0, self_loops=0):
    """src,dst ~ U[0,N) with `zero_in` nodes that never appear as destination,
    `dups` repeated edges and `self_loops` explicit self loops."""
    allowed = np.arange(N)
    if zero_in:
        banned = rng.choice(N, size=zero_in, replace=False)
        allowed = np.setdiff1d(allowed, banned)
    base = E - dups - self_loops
    src = rng.integers(0, N, size=base)
    dst = rng.choice(allowed, size=base)
    if dups:
        pick = rng.integers(0, base, size=dups)
        src = np.concatenate([src, src[pick]])
        dst = np.concatenate([dst, dst[pick]])
    if self_loops:
        sl = rng.choice(allowed, size=self_loops)
        src = np.concatenate([src, sl])
        dst = np.concatenate([dst, sl])
    perm = rng.permutation(src.shape[0])
    return np.stack([src[perm], dst[perm]]).astype(np.int64)


def run_single(ref, name, N, L, D, H, edge_index, seed, weight_edges=None, mha_cls=None):
    layer = make_layer(ref, D, H, seed, mha_cls)
    g = torch.Generator().manual_seed(seed + 1000)
    x = torch.randn(N, L * D, generator=g, requires_grad=True)
    dy = torch.randn(N, L * D, generator=g)
    ei = torch.from_numpy(edge_index)
    y = layer(x, ei)
    (y * dy).sum().backward()
    w = layer.attn_output_weights.detach().numpy()
    ao = layer.attn_output.detach().numpy()
    out = dict(N=N, L=L, D=D, H=H, edge_index=edge_index,
               x=x.detach().numpy(), dy=dy.numpy(), y=y.detach().numpy(), dx=x.grad.numpy())
    out.update(params_of(layer))
    out.update(grads_of(layer))
    if weight_edges is None:
        out['w_edges'] = np.arange(edge_index.shape[1], dtype=np.int64)
    else:
        out['w_edges'] = np.asarray(weight_edges, dtype=np.int64)
    out['attn_output_weights'] = w[out['w_edges']]
    out['attn_output'] = ao[out['w_edges'][:4]]          # a few per-edge outputs
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print(f'{name}: N={N} E={edge_index.shape[1]} L={L} D={D} H={H}')


def run_two_layer(ref, name, N, L, D, H, edge_index, seed):
    """conv -> ReLU -> conv, the call pattern of src/ampnet/module/amp_gcn.py:248-262."""
    l1 = make_layer(ref, D, H, seed)
    l2 = make_layer(ref, D, H, seed + 1)
    g = torch.Generator().manual_seed(seed + 1000)
    x = torch.randn(N, L * D, generator=g, requires_grad=True)
    dy = torch.randn(N, L * D, generator=g)
    ei = torch.from_numpy(edge_index)
    y = torch.relu(l2(torch.relu(l1(x, ei)), ei))
    (y * dy).sum().backward()
    out = dict(N=N, L=L, D=D, H=H, edge_index=edge_index,
               x=x.detach().numpy(), dy=dy.numpy(), y=y.detach().numpy(), dx=x.grad.numpy())
    out.update(params_of(l1, 'l1_'))
    out.update(params_of(l2, 'l2_'))
    out.update(grads_of(l1, 'l1_'))
    out.update(grads_of(l2, 'l2_'))
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print(f'{name}: two-layer N={N} E={edge_index.shape[1]} L={L} D={D} H={H}')


def main_linear(ref):
    """Fixtures of the softmax-free variant (file names linear_*.npz)."""
    mha = load_softmax_free_mha()
    rng = np.random.default_rng(20221005)
    toy_sl = np.array([[0, 1, 2, 3, 4], [2, 2, 2, 2, 2]], dtype=np.int64)
    run_single(ref, 'linear_toy5_selfloop_L2_D4_H2', 5, 2, 4, 2, toy_sl, seed=21, mha_cls=mha)
    run_single(ref, 'linear_cora_L20_D128_H4', 40, 20, 128, 4,
               random_graph(rng, 40, 260, zero_in=5, dups=16, self_loops=8), seed=22,
               weight_edges=np.arange(32), mha_cls=mha)
    run_single(ref, 'linear_wide_L4_D64_H8', 48, 4, 64, 8, random_graph(rng, 48, 350, zero_in=4), seed=23,
               mha_cls=mha)
    run_single(ref, 'linear_ampgcn_L40_D100_H2', 12, 40, 100, 2, random_graph(rng, 12, 60, zero_in=2),
               seed=24, weight_edges=np.arange(8), mha_cls=mha)


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    ref = load_reference()
    if '--linear-only' in sys.argv:            # add the softmax-free fixtures, leave the others as they are
        return main_linear(ref)
    rng = np.random.default_rng(20221004)

    # toy graph of synthetic_benchmark/testing_message_passing_pyg.py:24-33
    toy = np.array([[0, 1, 3, 4], [2, 2, 2, 2]], dtype=np.int64)
    toy_sl = np.array([[0, 1, 2, 3, 4], [2, 2, 2, 2, 2]], dtype=np.int64)
    run_single(ref, 'toy5_L1_D3_H1', 5, 1, 3, 1, toy, seed=1)
    run_single(ref, 'toy5_selfloop_L2_D4_H2', 5, 2, 4, 2, toy_sl, seed=2)
    # XOR model shape, synthetic_benchmark/xor_training_utils.py:58-72 (D=3, H=1, L=2)
    run_single(ref, 'xor_L2_D3_H1', 64, 2, 3, 1, random_graph(rng, 64, 256, zero_in=3), seed=3)
    # dense demo shape, examples/synthetic_benchmark.py:71-73 (D=3, H=1, L=100)
    run_single(ref, 'dense_L100_D3_H1', 24, 100, 3, 1, random_graph(rng, 24, 300, self_loops=10),
               seed=4, weight_edges=np.arange(8))
    # Cora harness shape, experiments/cora_benchmark_graphsaint.py:59-73 (D=128, H=4, L=20)
    cora = random_graph(rng, 40, 260, zero_in=5, dups=16, self_loops=8)
    run_single(ref, 'cora_L20_D128_H4', 40, 20, 128, 4, cora, seed=5)
    run_two_layer(ref, 'cora2layer_L20_D128_H4', 40, 20, 128, 4, cora, seed=6)
    # AMPGCN defaults, src/ampnet/module/amp_gcn.py:21-35 (D=100, H=2, L=40 -> dh=50)
    run_single(ref, 'ampgcn_L40_D100_H2', 16, 40, 100, 2, random_graph(rng, 16, 90, zero_in=2),
               seed=7, weight_edges=np.arange(8))
    # BASELINE config 3 shape (D=128, H=8 -> dh=16) and config 4 shape (D=256, H=8 -> dh=32)
    run_single(ref, 'cfg3_L20_D128_H8', 30, 20, 128, 8, random_graph(rng, 30, 200, zero_in=3, dups=8),
               seed=8, weight_edges=np.arange(32))
    run_single(ref, 'cfg4_L20_D256_H8', 20, 20, 256, 8, random_graph(rng, 20, 130, zero_in=2, dups=6, self_loops=4),
               seed=9, weight_edges=np.arange(32))
    run_single(ref, 'wide_L4_D64_H8', 48, 4, 64, 8, random_graph(rng, 48, 350, zero_in=4), seed=10)
    # hub: one destination with 2000 in-edges (segment-mean stress, BASELINE config 5)
    Nh = 700
    hub_src = rng.integers(0, Nh, size=2000)
    rest = random_graph(rng, Nh, 600, zero_in=20)
    hub = np.concatenate([np.stack([hub_src, np.full(2000, 17)]), rest], axis=1).astype(np.int64)
    hub = hub[:, rng.permutation(hub.shape[1])]
    run_single(ref, 'hub_L2_D8_H2', Nh, 2, 8, 2, hub, seed=11, weight_edges=np.arange(64))
    # L=1 degenerate (softmax over one element), BASELINE L=1 sweep
    run_single(ref, 'l1_L1_D128_H8', 80, 1, 128, 8, random_graph(rng, 80, 500, zero_in=6), seed=12)
    main_linear(ref)


if __name__ == '__main__':
    main()
