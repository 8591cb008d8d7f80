"""GPU parity: the HIP path (through the C ABI) against the reference's own
outputs (tests/golden) and against the CPU oracle on seeded inputs.

Tolerance (fp32): atol 1e-5, rtol 1e-4 on y, dx, weights; parameter
gradients are sums over N*L rows so their atol is scaled by max|grad|.  Rows with
in-degree 0 must be EXACTLY zero."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, golden_files, load_golden, assert_close_scaled

pytestmark = pytest.mark.gpu

SINGLE = golden_files(two_layer=False)
TWO = golden_files(two_layer=True)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from ampnet_amd import _lib
    _lib.load()                                     # the HIP extension must be the thing that runs
    return torch.device('cuda:0')


def _layer(g, dev, prefix='', gemm='native'):
    from ampnet_amd import AMPConv
    layer = AMPConv(int(g['D']), int(g['H'])).to(dev)
    layer.gemm_precision = gemm
    sd = {'multi_head_attention.in_proj_weight': torch.from_numpy(g[prefix + 'in_proj_weight']),
          'multi_head_attention.in_proj_bias': torch.from_numpy(g[prefix + 'in_proj_bias']),
          'multi_head_attention.out_proj.weight': torch.from_numpy(g[prefix + 'out_proj_weight']),
          'multi_head_attention.out_proj.bias': torch.from_numpy(g[prefix + 'out_proj_bias'])}
    layer.load_state_dict(sd)                       # reference checkpoints load by key
    return layer


def _grads(layer):
    m = layer.multi_head_attention
    return (m.in_proj_weight.grad.cpu().numpy(), m.in_proj_bias.grad.cpu().numpy(),
            m.out_proj.weight.grad.cpu().numpy(), m.out_proj.bias.grad.cpu().numpy())


def _check_single(g, dev, gemm='native'):
    layer = _layer(g, dev, gemm=gemm)
    x = torch.from_numpy(g['x']).to(dev).requires_grad_(True)
    ei = torch.from_numpy(g['edge_index']).to(dev)
    y = layer(x, ei)
    (y * torch.from_numpy(g['dy']).to(dev)).sum().backward()
    yh = y.detach().cpu().numpy()
    assert_close_scaled(yh, g['y'], 'y')
    deg = np.bincount(g['edge_index'][1], minlength=int(g['N']))
    assert (yh[deg == 0] == 0).all(), 'rows with no in-edge must be exactly 0'
    assert_close_scaled(x.grad.cpu().numpy(), g['dx'], 'dx')
    gw, gb, gow, gob = _grads(layer)
    assert_close_scaled(gw, g['g_in_proj_weight'], 'g_in_proj_weight')
    assert_close_scaled(gb, g['g_in_proj_bias'], 'g_in_proj_bias')
    assert_close_scaled(gow, g['g_out_proj_weight'], 'g_out_proj_weight')
    assert_close_scaled(gob, g['g_out_proj_bias'], 'g_out_proj_bias')
    w = layer.attn_output_weights
    assert tuple(w.shape) == (g['edge_index'].shape[1], int(g['L']), int(g['L']))
    wh = w.cpu().numpy()
    assert_close_scaled(wh[g['w_edges']], g['attn_output_weights'], 'attn_output_weights')
    np.testing.assert_allclose(wh.sum(-1), 1.0, atol=1e-5)
    ao = layer.attn_output.cpu().numpy()
    assert_close_scaled(ao[g['w_edges'][:4]], g['attn_output'], 'attn_output')
    return layer, x, ei


@pytest.mark.parametrize('path', SINGLE, ids=[os.path.basename(p)[:-4] for p in SINGLE])
def test_golden_single_layer(path, dev):
    # the shipped configuration (fp32 MFMA edge kernels, native projections) against the reference's outputs
    _check_single(load_golden(path), dev)


@pytest.mark.parametrize('path', SINGLE, ids=[os.path.basename(p)[:-4] for p in SINGLE])
def test_golden_single_layer_scaled_projections(path, dev, monkeypatch):
    # the fp32 projections' scaled two-plane mode (three fp16 products; what every large workload runs: functional.
    # PROJ_SCALED_MIN_ELEMENTS) forced onto the golden vectors' small graphs: same vectors, same tolerance
    from ampnet_amd.conv import functional as F_
    monkeypatch.setattr(F_, 'PROJ_SCALED_MIN_ELEMENTS', 0)
    # (every fixture with head width 32 and embed_dim % 128 == 0 -- the reference's own configuration, 128 / 4, among them --
    # also takes the plane-format edge passes of csrc/edge_mfma_f16x2.hip: tests/test_gpu_planes.py)
    calls = []
    real = F_.operand_stats
    monkeypatch.setattr(F_, 'operand_stats', lambda *a, **k: calls.append(1) or real(*a, **k))
    g = load_golden(path)
    _check_single(g, dev)
    assert calls or int(g['D']) % 4, 'the scaled mode did not run'


@pytest.mark.parametrize('path', SINGLE, ids=[os.path.basename(p)[:-4] for p in SINGLE])
def test_golden_gemm_library_fp32(path, dev):
    # the projections on the library's fp32 GEMMs (gemm_precision='fp32': what serves bf16 storage and
    # embed_dim % 128 != 0, and round 2's only path) against the same vectors; the default 'native' mode
    # (csrc/proj_gemm.hip) is what every other golden test runs where the shape allows
    _check_single(load_golden(path), dev, gemm='fp32')


@pytest.mark.parametrize('path', SINGLE, ids=[os.path.basename(p)[:-4] for p in SINGLE])
def test_golden_gemm_bf16x3(path, dev):
    # projections on hipBLASLt's 3-product bf16 split (functional.gemm_precision): same fp32
    # tolerance against the reference's outputs, and torch's global switches come back as they were
    before = (torch.backends.cuda.matmul.allow_tf32, torch.backends.cuda.preferred_blas_library(),
              os.environ.get('HIPBLASLT_ALLOW_TF32'))
    _check_single(load_golden(path), dev, gemm='bf16x3')
    after = (torch.backends.cuda.matmul.allow_tf32, torch.backends.cuda.preferred_blas_library(),
             os.environ.get('HIPBLASLT_ALLOW_TF32'))
    assert before == after


def test_gemm_precision_rejects_unknown_mode(dev):
    from ampnet_amd.conv.functional import gemm_precision
    with pytest.raises(ValueError):
        with gemm_precision('fp16'):
            pass


@pytest.mark.parametrize('path', SINGLE[:4], ids=[os.path.basename(p)[:-4] for p in SINGLE[:4]])
def test_golden_generic_kernels(path, dev, monkeypatch):
    # the shape-generic kernels must agree with the reference on every shape too
    monkeypatch.setenv('AMPCONV_FORCE_GENERIC', '1')
    _check_single(load_golden(path), dev)


@pytest.mark.parametrize('path', TWO, ids=[os.path.basename(p)[:-4] for p in TWO])
def test_golden_two_layer(path, dev):
    # conv -> ReLU -> conv -> ReLU, src/ampnet/module/amp_gcn.py:248-262
    g = load_golden(path)
    l1, l2 = _layer(g, dev, 'l1_'), _layer(g, dev, 'l2_')
    x = torch.from_numpy(g['x']).to(dev).requires_grad_(True)
    ei = torch.from_numpy(g['edge_index']).to(dev)
    y = torch.relu(l2(torch.relu(l1(x, ei)), ei))
    (y * torch.from_numpy(g['dy']).to(dev)).sum().backward()
    assert_close_scaled(y.detach().cpu().numpy(), g['y'], 'y')
    assert_close_scaled(x.grad.cpu().numpy(), g['dx'], 'dx')
    for p, layer in (('l1_', l1), ('l2_', l2)):
        gw, gb, gow, gob = _grads(layer)
        assert_close_scaled(gw, g[p + 'g_in_proj_weight'], p + 'g_in_proj_weight')
        assert_close_scaled(gb, g[p + 'g_in_proj_bias'], p + 'g_in_proj_bias')
        assert_close_scaled(gow, g[p + 'g_out_proj_weight'], p + 'g_out_proj_weight')
        assert_close_scaled(gob, g[p + 'g_out_proj_bias'], p + 'g_out_proj_bias')


def test_message_matches_per_edge_reference(dev):
    # message(x_i, x_j) on pre-gathered rows == the reference's per-edge attn_output (amp_conv.py:39,49)
    g = load_golden([p for p in SINGLE if 'cora_L20' in p][0])
    layer = _layer(g, dev)
    x = torch.from_numpy(g['x']).to(dev)
    ei = torch.from_numpy(g['edge_index']).to(dev)
    sel = torch.from_numpy(g['w_edges'][:4]).to(dev)
    m = layer.message(x[ei[1, sel]], x[ei[0, sel]])
    L, D = int(g['L']), int(g['D'])
    assert_close_scaled(m.detach().cpu().numpy().reshape(4, L, D), g['attn_output'], 'message')
    assert_close_scaled(layer.attn_output_weights.cpu().numpy(), g['attn_output_weights'][:4], 'w')


def test_aggregate_known_answer(dev):
    # synthetic_benchmark/testing_message_passing_pyg.py:23-40
    from ampnet_amd import AMPConv
    layer = AMPConv(3, 1).to(dev)
    x = torch.tensor([[1, 1, 1], [2, 2, 2], [3, 3, 3], [10, 10, 10], [11, 11, 11]],
                     dtype=torch.float32, device=dev)
    for ei, want in (([[0, 1, 3, 4], [2, 2, 2, 2]], 6.0), ([[0, 1, 2, 3, 4], [2, 2, 2, 2, 2]], 5.4)):
        ei = torch.tensor(ei, device=dev)
        out = layer.aggregate(x[ei[0]], ei[1], dim_size=5).cpu().numpy()
        np.testing.assert_allclose(out[2], [want] * 3, rtol=1e-6)
        assert (out[[0, 1, 3, 4]] == 0).all()


@pytest.mark.parametrize('N,E,hub', [(1000, 20000, 4000), (2708, 10556, 300), (16384, 12288, 0), (37, 65, 0), (500, 12289, 70)],
                         ids=['large-path', 'cora-small-path', 'small-path-limits', 'tiny', 'just-over-the-limit'])
def test_csr_build_is_a_stable_sort(N, E, hub, dev):
    """Both preparation paths of ampconv_graph_build -- one launch for graphs of at most 12 288 edges / 16 384 nodes
    (the stable sorts in LDS), device-wide radix sorts above -- against numpy's stable argsort: positions, pointers,
    1 / in-degree, and the long-segment plans as SETS of descriptors (their slot order is arbitrary by design)."""
    from ampnet_amd import EdgeCSR, _lib
    rng = np.random.default_rng(3)
    ei = rng.integers(0, N, size=(2, E)).astype(np.int64)
    ei[1, :hub] = 7                                   # a hub destination ...
    ei[0, hub:hub + hub // 2] = 11                    # ... and a hub source
    csr = EdgeCSR(torch.from_numpy(ei).to(dev), N)
    perm = np.argsort(ei[1], kind='stable')
    np.testing.assert_array_equal(csr.eperm.cpu().numpy(), perm)
    np.testing.assert_array_equal(csr.col.cpu().numpy(), ei[0][perm])
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei[1], minlength=N))])
    np.testing.assert_array_equal(csr.rowptr.cpu().numpy(), rowptr)
    cperm = np.argsort(ei[0], kind='stable')
    np.testing.assert_array_equal(csr.cperm.cpu().numpy(), cperm)
    np.testing.assert_array_equal(csr.crow.cpu().numpy(), ei[1][cperm])
    cscptr = np.concatenate([[0], np.cumsum(np.bincount(ei[0], minlength=N))])
    np.testing.assert_array_equal(csr.cscptr.cpu().numpy(), cscptr)
    np.testing.assert_array_equal(csr.cinv.cpu().numpy(), (1.0 / np.bincount(ei[1], minlength=N)[ei[1][cperm]]).astype(np.float32))
    chunk = csr.hub_chunk
    assert chunk == _lib.hub_chunk(E)
    for plan, n, ptr in ((csr.hub_dst, csr.hub_dst_chunks, rowptr), (csr.hub_src, csr.hub_src_chunks, cscptr)):
        want = set()
        for r in np.nonzero(np.diff(ptr) > chunk)[0]:
            b, e = int(ptr[r]), int(ptr[r + 1])
            nc = -(-(e - b) // chunk)
            want |= {(int(r), b + k * chunk, min(e, b + (k + 1) * chunk), nc if k == 0 else 0) for k in range(nc)}
        assert n == len(want)
        if n:
            got = plan.cpu().numpy()[4:4 + 4 * n].reshape(n, 4)
            assert {tuple(int(v) for v in row) for row in got} == want


def test_small_graph_preparation_flags_out_of_range_ids(dev):
    from ampnet_amd import EdgeCSR
    ei = torch.tensor([[0, 1, 2, 9], [1, 2, 0, 1]], device=dev)
    with pytest.raises(ValueError, match='outside'):
        EdgeCSR(ei, 5)
    ei[0, 3] = -1
    with pytest.raises(ValueError, match='outside'):
        EdgeCSR(ei, 5)


def test_edge_cases(dev):
    from ampnet_amd import AMPConv
    layer = AMPConv(8, 2).to(dev)
    x = torch.randn(6, 16, device=dev, requires_grad=True)
    # empty edge set: every row is exactly zero, gradients flow as zeros
    y = layer(x, torch.zeros(2, 0, dtype=torch.int64, device=dev))
    assert (y == 0).all()
    y.sum().backward()
    assert (x.grad == 0).all()
    assert tuple(layer.attn_output_weights.shape) == (0, 2, 2)
    # out-of-range node id is rejected before any kernel touches it
    with pytest.raises(ValueError, match='outside'):
        layer(x, torch.tensor([[0, 9], [1, 2]], device=dev))
    with pytest.raises(ValueError, match='int64'):
        layer(x, torch.tensor([[0], [1]], dtype=torch.int32, device=dev))


@pytest.mark.parametrize('shape', [(3000, 30000, 20, 128, 4), (2000, 24000, 20, 256, 8),
                                   (1500, 9000, 20, 128, 8), (500, 4000, 7, 24, 3),
                                   (800, 6000, 13, 64, 2), (800, 6000, 17, 64, 4), (600, 5000, 3, 32, 2)],
                         ids=['cora_like', 'cfg4_like', 'cfg3_like', 'odd', 'L13_dh32', 'L17_dh16', 'L3_dh16'])
@pytest.mark.parametrize('scaled', [False, True], ids=['six_products', 'scaled_planes'])
def test_seeded_vs_oracle(shape, scaled, dev, monkeypatch):
    """Larger seeded graphs (uniform + one hub + isolated nodes) against the numpy oracle, the fp32 projections in
    both forms (six bf16 products: what graphs of this size run by default; scaled two-plane: what large ones run)."""
    from ampnet_amd import AMPConv
    from ampnet_amd.conv import functional as F_
    from oracle.ampconv_numpy import AMPConvOracle
    monkeypatch.setattr(F_, 'PROJ_SCALED_MIN_ELEMENTS', 0 if scaled else 1 << 62)
    N, E, L, D, H = shape
    torch.manual_seed(11)
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(N, L * D, generator=g)
    dy = torch.randn(N, L * D, generator=g)
    ei = torch.randint(0, N - 50, (2, E), generator=g)       # last 50 nodes isolated
    ei[1, : E // 10] = 3                                      # hub: 10 % of the edges end at node 3
    ei[0, E // 10: E // 5] = 7                                # ... and 10 % start at node 7 (CSC hub)
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg, ei.to(dev))
    (y * dy.to(dev)).sum().backward()
    m = layer.multi_head_attention
    o = AMPConvOracle(m.in_proj_weight.detach().cpu().numpy(), m.in_proj_bias.detach().cpu().numpy(),
                      m.out_proj.weight.detach().cpu().numpy(), m.out_proj.bias.detach().cpu().numpy(),
                      H, dtype=np.float64, edge_chunk=2048)
    y_ref, _ = o.forward(x.numpy(), ei.numpy(), need_weights=False)
    dx_ref, dWin, dbin, dWo, dbo = o.backward(dy.numpy())
    yh = y.detach().cpu().numpy()
    assert_close_scaled(yh, y_ref, 'y')
    assert (yh[N - 50:] == 0).all()
    assert_close_scaled(xg.grad.cpu().numpy(), dx_ref, 'dx')
    gw, gb, gow, gob = _grads(layer)
    assert_close_scaled(gw, dWin, 'g_in_proj_weight')
    assert_close_scaled(gb, dbin, 'g_in_proj_bias')
    assert_close_scaled(gow, dWo, 'g_out_proj_weight')
    assert_close_scaled(gob, dbo, 'g_out_proj_bias')


def test_properties_medium(dev):
    """Size-independent properties: bitwise run-to-run determinism, invariance of the
    mean to duplicating every edge, softmax rows summing to 1."""
    from ampnet_amd import AMPConv
    torch.manual_seed(5)
    N, E, L, D, H = 20000, 200000, 20, 128, 8
    layer = AMPConv(D, H).to(dev)
    layer.retain_attention = False
    x = torch.randn(N, L * D, device=dev)
    ei = torch.randint(0, N, (2, E), device=dev)
    with torch.no_grad():
        y1 = layer(x, ei)
        y2 = layer(x, ei.clone())
        assert torch.equal(y1, y2)
        y3 = layer(x, torch.cat([ei, ei], dim=1))
        torch.testing.assert_close(y3, y1, rtol=1e-4, atol=1e-5)
    layer.retain_attention = True
    with torch.no_grad():
        layer(x[:2000], ei[:, :5000] % 2000)
        w = layer.attn_output_weights
        torch.testing.assert_close(w.sum(-1), torch.ones_like(w[..., 0]), rtol=0, atol=1e-5)


def _oracle_rows(layer, x, ei, rows, H):
    """Oracle forward for a few destination rows of a big graph: the sub-problem made of the
    in-edges of `rows` (sources relabelled) reproduces those rows exactly."""
    from oracle.ampconv_numpy import AMPConvOracle
    src, dst = ei[0].cpu().numpy(), ei[1].cpu().numpy()
    sel = np.isin(dst, rows)
    s_sub, d_sub = src[sel], dst[sel]
    nodes = np.unique(np.concatenate([rows, s_sub]))
    relabel = {int(n): i for i, n in enumerate(nodes)}
    ei_sub = np.array([[relabel[int(s)] for s in s_sub], [relabel[int(d)] for d in d_sub]], dtype=np.int64)
    m = layer.multi_head_attention
    o = AMPConvOracle(m.in_proj_weight.detach().cpu().numpy(), m.in_proj_bias.detach().cpu().numpy(),
                      m.out_proj.weight.detach().cpu().numpy(), m.out_proj.bias.detach().cpu().numpy(),
                      H, dtype=np.float64)
    y_sub, _ = o.forward(x[torch.from_numpy(nodes).to(x.device)].cpu().numpy(), ei_sub, need_weights=False)
    return y_sub[[relabel[int(r)] for r in rows]]


def test_full_size_config3(dev, monkeypatch):
    """BASELINE config 3 at full size (100k nodes / 1M edges, L=20, D=128, H=8):
    sampled destination rows against the oracle, MFMA kernels against the independent
    shape-generic kernels (forward and every gradient), run-to-run bitwise determinism."""
    from ampnet_amd import AMPConv, graph_cache
    N, E, L, D, H = 100_000, 1_000_000, 20, 128, 8
    torch.manual_seed(21)
    layer = AMPConv(D, H).to(dev)
    layer.retain_attention = False
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    g = torch.Generator(device=dev).manual_seed(22)
    x = torch.randn(N, L * D, device=dev, generator=g)
    dy = torch.randn(N, L * D, device=dev, generator=g)
    ei = torch.randint(0, N, (2, E), device=dev, generator=g)
    ei[1, :3000] = 77                                         # a 3000-in-edge hub
    ei[0, 3000:5000] = 78                                     # a 2000-out-edge hub

    def run():
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y = layer(xg, ei)
        y.backward(dy)
        m = layer.multi_head_attention
        return [t.detach().clone() for t in (y, xg.grad, m.in_proj_weight.grad, m.in_proj_bias.grad,
                                             m.out_proj.weight.grad, m.out_proj.bias.grad)]

    fast = run()
    again = run()
    for a, b in zip(fast, again):
        assert torch.equal(a, b), 'not bitwise reproducible'
    rows = np.array([77, 0, 1, 5, 4242, 99_999, 31_337, 12_345])
    y_ref = _oracle_rows(layer, x, ei, rows, H)
    assert_close_scaled(fast[0][torch.from_numpy(rows).to(dev)].cpu().numpy(), y_ref, 'y[sampled rows]')
    monkeypatch.setenv('AMPCONV_FORCE_GENERIC', '1')
    slow = run()
    names = ['y', 'dx', 'g_in_proj_weight', 'g_in_proj_bias', 'g_out_proj_weight', 'g_out_proj_bias']
    for n, a, b in zip(names, fast, slow):
        assert_close_scaled(a.cpu().numpy(), b.cpu().numpy(), n + ' (mfma vs generic)')
    deg = torch.bincount(ei[1], minlength=N)
    assert (fast[0][deg == 0] == 0).all()


def test_edge_kernels_error_vs_fp64_oracle(dev):
    """The fp32 layer against an fp64 oracle at a BASELINE-shaped size: the error stays far inside the stated
    tolerance (max error / max entry; DESIGN.md section 4a has the table)."""
    from ampnet_amd import AMPConv
    from oracle.ampconv_numpy import AMPConvOracle
    N, E, L, D, H = 1500, 15000, 20, 256, 8
    torch.manual_seed(31)
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(32)
    x = torch.randn(N, L * D, generator=g) * 2.0
    dy = torch.randn(N, L * D, generator=g)
    ei = torch.randint(0, N, (2, E), generator=g)
    m = layer.multi_head_attention
    o = AMPConvOracle(m.in_proj_weight.detach().cpu().numpy(), m.in_proj_bias.detach().cpu().numpy(),
                      m.out_proj.weight.detach().cpu().numpy(), m.out_proj.bias.detach().cpu().numpy(),
                      H, dtype=np.float64, edge_chunk=2048)
    y_ref, _ = o.forward(x.numpy(), ei.numpy(), need_weights=False)
    dx_ref = o.backward(dy.numpy())[0]
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg, ei.to(dev))
    y.backward(dy.to(dev))
    err = (np.abs(y.detach().cpu().numpy() - y_ref).max() / np.abs(y_ref).max(),
           np.abs(xg.grad.cpu().numpy() - dx_ref).max() / np.abs(dx_ref).max())
    print('max abs error / max |ref| (y, dx):', err)
    assert err[0] < 5e-6 and err[1] < 5e-6


def _ddp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)   # both ranks share cuda:0 here;
    from ampnet_amd import AMPConv                                   # on the 8-GPU node: nccl = RCCL
    from ampnet_amd.distributed import GradientAllReducer, broadcast_parameters
    dev = torch.device('cuda:0')
    torch.manual_seed(100 + rank)                                    # different init per rank ...
    layer = AMPConv(128, 4).to(dev)
    broadcast_parameters(layer, src=0)                               # ... made equal here
    reducer = GradientAllReducer(layer.parameters())
    opt = torch.optim.Adam(layer.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(7 + rank)                      # one subgraph per rank
    N, E, L = 300, 2500, 20
    for step in range(2):
        x = torch.randn(N, L * 128, generator=g).to(dev)
        ei = torch.randint(0, N, (2, E), generator=g).to(dev)
        opt.zero_grad()
        layer(x, ei).pow(2).mean().backward()
        if step == 0:
            local = [p.grad.detach().cpu().clone() for p in layer.parameters()]
        reducer.allreduce()
        if step == 0:
            avg = [p.grad.detach().cpu().clone() for p in layer.parameters()]
        opt.step()
    torch.save({'params': [p.detach().cpu() for p in layer.parameters()], 'local': local, 'avg': avg},
               os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_data_parallel_step_two_ranks(dev, tmp_path):
    """One-subgraph-per-rank data parallelism with the HIP layer (2 processes): averaged gradient
    = mean of the per-rank gradients, identical parameters on both ranks after optimizer steps
    (experiments/cora_benchmark_graphsaint_distributed.py:63-94 as intended)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f'r{i}.pt')) for i in range(2)]
    for a, b in zip(r[0]['params'], r[1]['params']):
        assert torch.equal(a, b)
    for l0, l1, a0, a1 in zip(r[0]['local'], r[1]['local'], r[0]['avg'], r[1]['avg']):
        torch.testing.assert_close(a0, (l0 + l1) / 2, rtol=1e-5, atol=1e-7)
        assert torch.equal(a0, a1)


@pytest.mark.parametrize('shape', [(1200, 12000, 20, 256, 8), (700, 5000, 13, 64, 2), (900, 9000, 20, 128, 8),
                                   (500, 4000, 7, 48, 3), (300, 2500, 40, 100, 2), (260, 2000, 33, 24, 2)],
                         ids=['L20_D256', 'L13_D64', 'L20_D128_dh16', 'L7_D48_dh16', 'L40_D100_dh50_block',
                              'L33_D24_dh12_block'])
def test_bf16_storage(shape, dev):
    """bf16-storage mode (BASELINE config 5: Q/K/V/O and gradients in bf16, fp32 accumulate; head width 32, and
    16 -- BASELINE config 3's -- as half-filled tiles; the reference's class default L=40, D=100, H=2 and other even
    head widths on the workgroup-per-unit kernels, which widen / round the rows themselves)
    against the fp64 oracle evaluated on the same bf16-rounded inputs and parameters.
    Tolerance: rtol 2e-2, atol 2e-2 of the tensor's scale."""
    from ampnet_amd import AMPConv
    from oracle.ampconv_numpy import AMPConvOracle
    N, E, L, D, H = shape
    torch.manual_seed(41)
    layer = AMPConv(D, H)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    layer = layer.to(dev).to(torch.bfloat16)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(N, L * D, generator=g).to(torch.bfloat16)
    dy = torch.randn(N, L * D, generator=g).to(torch.bfloat16)
    ei = torch.randint(0, N - 20, (2, E), generator=g)
    ei[1, : E // 10] = 3                                     # dst hub (long CSR segment)
    ei[0, E // 10: E // 5] = 7                               # src hub (long CSC segment)
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg, ei.to(dev))
    assert y.dtype == torch.bfloat16
    y.backward(dy.to(dev))
    m = layer.multi_head_attention
    f64 = lambda t: t.detach().float().cpu().numpy().astype(np.float64)
    o = AMPConvOracle(f64(m.in_proj_weight), f64(m.in_proj_bias), f64(m.out_proj.weight), f64(m.out_proj.bias),
                      H, dtype=np.float64, edge_chunk=2048)
    y_ref, w_ref = o.forward(f64(x), ei.numpy(), need_weights=True)
    dx_ref, dWin, dbin, dWo, dbo = o.backward(f64(dy))
    assert_close_scaled(f64(y), y_ref, 'y', atol=2e-2, rtol=2e-2)
    assert (f64(y)[N - 20:] == 0).all()
    assert_close_scaled(f64(xg.grad), dx_ref, 'dx', atol=2e-2, rtol=2e-2)
    assert_close_scaled(f64(m.in_proj_weight.grad), dWin, 'g_in_proj_weight', atol=2e-2, rtol=2e-2)
    assert_close_scaled(f64(m.in_proj_bias.grad), dbin, 'g_in_proj_bias', atol=2e-2, rtol=2e-2)
    assert_close_scaled(f64(m.out_proj.weight.grad), dWo, 'g_out_proj_weight', atol=2e-2, rtol=2e-2)
    assert_close_scaled(f64(m.out_proj.bias.grad), dbo, 'g_out_proj_bias', atol=2e-2, rtol=2e-2)
    w = layer.attn_output_weights.cpu().numpy()
    assert_close_scaled(w, w_ref, 'attn_output_weights', atol=2e-2, rtol=2e-2)


def test_bf16_storage_rejects_other_head_widths(dev):
    from ampnet_amd import AMPConv
    layer = AMPConv(66, 2).to(dev).to(torch.bfloat16)                   # dh = 33: odd, fp32 only (edge_generic.hip)
    x = torch.randn(10, 4 * 66, device=dev).to(torch.bfloat16)
    ei = torch.randint(0, 10, (2, 30), device=dev)
    with pytest.raises(ValueError, match='head dimensions 32 and 16'):
        layer(x, ei)


def test_training_steps_match_cpu_reference(dev):
    """Three SGD steps of a 2-layer conv -> ReLU -> conv -> ReLU -> token-mean -> Linear model
    (the AMPGCN call pattern, src/ampnet/module/amp_gcn.py:248-274) on the GPU path and on the
    reference-shaped CPU restatement, same initial parameters and batches: losses and final
    parameters must track each other."""
    from ampnet_amd import AMPConv
    from oracle.ampconv_torch import RefShapedAMPConv
    N, E, L, D, H, C = 300, 2400, 20, 128, 4, 7

    class Net(torch.nn.Module):
        def __init__(self, conv_cls):
            super().__init__()
            self.conv1, self.conv2 = conv_cls(D, H), conv_cls(D, H)
            self.out = torch.nn.Linear(D, C)

        def forward(self, x, ei):
            x = torch.relu(self.conv1(x, ei))
            x = torch.relu(self.conv2(x, ei))
            return torch.log_softmax(self.out(x.reshape(x.shape[0], L, D).mean(dim=1)), dim=1)

    torch.manual_seed(3)
    ref = Net(RefShapedAMPConv)
    gpu = Net(AMPConv)
    gpu.load_state_dict(ref.state_dict())            # identical state-dict keys
    gpu = gpu.to(dev)
    # plain SGD: parameter differences stay proportional to gradient differences (Adam turns
    # round-off-level gradients into +-lr steps, which is not a property of the layer)
    opt_r = torch.optim.SGD(ref.parameters(), lr=0.5)
    opt_g = torch.optim.SGD(gpu.parameters(), lr=0.5)
    g = torch.Generator().manual_seed(4)
    for step in range(3):
        x = torch.randn(N, L * D, generator=g)
        ei = torch.randint(0, N, (2, E), generator=g)
        yl = torch.randint(0, C, (N,), generator=g)
        opt_r.zero_grad(); opt_g.zero_grad()
        loss_r = torch.nn.functional.nll_loss(ref(x, ei), yl)
        loss_g = torch.nn.functional.nll_loss(gpu(x.to(dev), ei.to(dev)), yl.to(dev))
        loss_r.backward(); loss_g.backward()
        opt_r.step(); opt_g.step()
        assert abs(loss_r.item() - loss_g.item()) < 1e-4 * max(1.0, abs(loss_r.item())), (step, loss_r.item(), loss_g.item())
    for (n, a), b in zip(ref.named_parameters(), gpu.parameters()):
        torch.testing.assert_close(b.detach().cpu(), a.detach(), rtol=1e-3, atol=2e-5, msg=lambda m: f'{n}: {m}')


@pytest.mark.parametrize('family', ['wave', 'block'])
@pytest.mark.parametrize('seed', range(10))
def test_random_shapes_mfma_vs_generic(seed, family, dev, monkeypatch):
    """Random small problems, degenerate graphs included: the MFMA kernels and the independent
    shape-generic kernels must agree on y and every gradient.  'wave' = edge_mfma.hip shapes
    (L in 1..20, dh in {16, 32}), 'block' = the workgroup-per-unit shapes of edge_block_x3.hip (L in 1..64, even dh <= 64,
    e.g. the reference's class defaults L = 40, dh = 50)."""
    from ampnet_amd import AMPConv, graph_cache
    rng = np.random.default_rng(1000 + seed + (500 if family == 'block' else 0))
    if family == 'wave':
        dh = int(rng.choice([16, 32]))
        H = int(rng.choice([1, 2, 3, 4, 8, 16]))
        L = int(rng.integers(1, 21))
    else:
        dh = int(rng.choice([2, 6, 10, 20, 34, 48, 50, 64]))
        H = int(rng.choice([1, 2, 3, 4]))
        L = int(rng.integers(1, 65)) if seed % 2 else int(rng.integers(21, 65))
    D = dh * H
    N = int(rng.integers(1, 400))
    kind = seed % 5
    E = int(rng.integers(0, 6 * N + 2))
    src = rng.integers(0, N, E)
    dst = rng.integers(0, N, E)
    if kind == 1:
        dst[:] = rng.integers(0, N)                       # every edge into one node (one long segment)
    elif kind == 2:
        src[:] = rng.integers(0, N)                       # every edge out of one node
    elif kind == 3:
        dst = src.copy()                                  # self loops only
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(dev)
    torch.manual_seed(seed)
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.2)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.2)
    x = torch.randn(N, L * D, device=dev)
    dy = torch.randn(N, L * D, device=dev)

    def run():
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y = layer(xg, ei)
        y.backward(dy)
        m = layer.multi_head_attention
        return [t.detach().cpu().numpy() for t in (y, xg.grad, m.in_proj_weight.grad, m.in_proj_bias.grad,
                                                    m.out_proj.weight.grad, m.out_proj.bias.grad)]

    fast = run()
    monkeypatch.setenv('AMPCONV_FORCE_GENERIC', '1')
    slow = run()
    for name, a, b in zip(['y', 'dx', 'gWin', 'gbin', 'gWo', 'gbo'], fast, slow):
        # kind 2: every edge leaves ONE node, whose dx row is a sum over all E edges -- an accumulated quantity like the
        # parameter gradients (two fp32 summation orders differ by ~1e-6 of its magnitude, 2-3e-5 absolute here)
        assert_close_scaled(a, b, f'{name} (N={N} E={E} L={L} D={D} H={H} kind={kind})',
                            scaled=True if (name == 'dx' and kind == 2) else None)


@pytest.mark.parametrize('shape', [(900, 9000, 20, 256, 8), (500, 6000, 13, 64, 4), (400, 5000, 20, 64, 4),
                                   (350, 4000, 18, 128, 4), (350, 4000, 17, 32, 2)],
                         ids=['L20_dh32', 'L13_dh16', 'L20_dh16_hub', 'L18_dh32', 'L17_dh16'])
def test_softmax_stats_handoff_matches_own_reduction(shape, dev, monkeypatch):
    # the source pass either re-reduces softmax / delta across lanes or reads what the destination
    # pass stored (include/ampconv.h "Softmax statistics"): same gradients to fp32 rounding, on
    # ragged graphs with hubs (long segments are cut into chunks) and isolated nodes
    from ampnet_amd import AMPConv, graph_cache, _lib
    from ampnet_amd.conv import functional as F_
    N, E, L, D, H = shape
    tdt, code = torch.float32, _lib.AMPCONV_F32
    g = torch.Generator().manual_seed(E)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, : E // 10] = 3                            # hub destination (in-degree >> chunk)
    ei[0, E // 10: E // 5] = 5                      # hub source
    ei[1, ei[1] == 7] = 8                           # node 7 receives nothing
    x = torch.randn(N, L * D, generator=g)
    dy = torch.randn(N, L * D, generator=g)
    torch.manual_seed(3)
    layer = AMPConv(D, H).to(dev).to(tdt)
    assert _lib.load().ampconv_softmax_stats_bytes(E, L, D, H, code) == E * H * 40 * 4
    res = []
    for on in (True, False):
        monkeypatch.setattr(F_, 'SOFTMAX_STATS', on)
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xg = x.to(dev).to(tdt).requires_grad_(True)
        layer(xg, ei.to(dev)).backward(dy.to(dev).to(tdt))
        m = layer.multi_head_attention
        res.append([t.float().cpu().numpy() for t in (xg.grad, m.in_proj_weight.grad, m.in_proj_bias.grad,
                                                      m.out_proj.weight.grad, m.out_proj.bias.grad)])
    rtol, atol = 2e-5, 2e-6                        # the only difference is the rounding of lse
    for a, b, name in zip(res[0], res[1], ('dx', 'gW_in', 'gb_in', 'gW_out', 'gb_out')):
        scale = max(1.0, float(np.abs(b).max()))
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=name)


def test_softmax_stats_rejected_where_unsupported(dev):
    # shapes on the generic kernels keep no statistics: size query says 0, passing a buffer is an error
    from ampnet_amd import _lib
    lib = _lib.load()
    assert lib.ampconv_softmax_stats_bytes(1000, 70, 100, 2, _lib.AMPCONV_F32) == 0       # L > 64
    assert lib.ampconv_softmax_stats_bytes(1000, 2, 3, 1, _lib.AMPCONV_F32) == 0          # odd dh
    assert lib.ampconv_softmax_stats_bytes(1000, 40, 100, 2, _lib.AMPCONV_F32) == 1000 * 2 * 2 * 48 * 4   # edge_block_x3.hip
    assert lib.ampconv_softmax_stats_bytes(1000, 20, 256, 8, _lib.AMPCONV_BF16) == 0      # HBM-bound: no gain


@pytest.mark.parametrize('shape', [(260, 2600, 17, 64, 2), (260, 2600, 18, 128, 8), (260, 2600, 19, 96, 3),
                                   (260, 2600, 20, 32, 2)], ids=['L17_dh32', 'L18_dh16', 'L19_dh32', 'L20_dh16'])
def test_batched_tail_kernels_vs_generic(shape, dev, monkeypatch):
    """The batched-tail kernels (fwd / dst / src `_t4`: tail tokens of four edges in one MFMA tile) on
    partially filled tails (L = 17..19), segment lengths of every residue mod 4, a hub and isolated
    nodes, against the independent shape-generic kernels; and with the batching switched off."""
    from ampnet_amd import AMPConv, graph_cache
    N, E, L, D, H = shape
    g = torch.Generator().manual_seed(L * 1000 + D)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :150] = 3                                   # hub destination
    ei[0, 150:300] = 5                                # hub source
    for k in range(1, 8):                             # nodes with exactly k in-edges and k out-edges
        ei[1, 300 + 10 * k: 300 + 11 * k] = 100 + k
        ei[1, ei[1] == 100 + k] = 100 + k
        ei[0, 600 + 10 * k: 600 + 11 * k] = 200 + k
    ei[1, ei[1] == 7] = 8                             # node 7 receives nothing
    torch.manual_seed(L)
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.2)
    x = torch.randn(N, L * D, generator=g).to(dev)
    dy = torch.randn(N, L * D, generator=g).to(dev)

    def run():
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y = layer(xg, ei.to(dev))
        y.backward(dy)
        m = layer.multi_head_attention
        return [t.detach().cpu().numpy() for t in (y, xg.grad, m.in_proj_weight.grad, m.in_proj_bias.grad,
                                                    m.out_proj.weight.grad, m.out_proj.bias.grad)]

    fast = run()
    monkeypatch.setenv('AMPCONV_FORCE_GENERIC', '1')
    slow = run()
    for name, a, b in zip(['y', 'dx', 'gWin', 'gbin', 'gWo', 'gbo'], fast, slow):
        assert_close_scaled(a, b, name)


def test_batched_tail_kernels_large_scores(dev, monkeypatch):
    """Scores spread over hundreds of log2 units between the edges of one destination: in the
    batched-tail kernels every lane group evaluates exp2(tail score of ITS edge - max of the edge being
    processed), which overflows to inf in the groups that are then discarded by a select.  No inf or
    NaN may leak into the results, and they must still agree with the generic kernels."""
    from ampnet_amd import AMPConv, graph_cache
    N, E, L, D, H = 200, 2400, 20, 64, 2
    g = torch.Generator().manual_seed(99)
    ei = torch.randint(0, N, (2, E), generator=g).to(dev)
    torch.manual_seed(17)
    layer = AMPConv(D, H).to(dev)
    x = torch.randn(N, L * D, generator=g)
    x[::3] *= 25.0                                  # a third of the nodes produce huge keys / queries
    x = x.to(dev)
    dy = torch.randn(N, L * D, generator=g).to(dev)

    def run():
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y = layer(xg, ei)
        y.backward(dy)
        m = layer.multi_head_attention
        return [t.detach().cpu().numpy() for t in (y, xg.grad, m.in_proj_weight.grad, m.out_proj.weight.grad)]

    fast = run()
    for t in fast:
        assert np.isfinite(t).all()
    monkeypatch.setenv('AMPCONV_FORCE_GENERIC', '1')
    slow = run()
    for name, a, b in zip(['y', 'dx', 'gWin', 'gWo'], fast, slow):
        # saturated softmaxes amplify rounding: scores of magnitude ~500 carry an absolute fp32 error of ~3e-5, which is
        # the RELATIVE error of the probabilities next to a tie; two implementations that round the scores differently
        # differ by that much of the output's magnitude -- magnitude-scaled tolerance for this cross-check
        assert_close_scaled(a, b, name, atol=3e-5, rtol=3e-4, scaled=True)


@pytest.mark.parametrize('shape', [(300, 3000, 40, 100, 2), (300, 3000, 24, 128, 2)], ids=['L40_dh50', 'L24_dh64'])
def test_block_kernels_long_segments(shape, dev, monkeypatch):
    """Workgroup-per-unit kernels (edge_block_x3.hip) on a graph with a hub destination and a hub source
    (segments of ~600 edges, cut into 64-edge chunks -> partial tiles -> ordered combine) against the
    shape-generic kernels, which walk every segment in one piece."""
    from ampnet_amd import AMPConv, graph_cache
    N, E, L, D, H = shape
    g = torch.Generator().manual_seed(L + D)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :600] = 3
    ei[0, 600:1200] = 5
    ei[1, ei[1] == 7] = 8
    ei = ei.to(dev)
    torch.manual_seed(2)
    layer = AMPConv(D, H).to(dev)
    x = torch.randn(N, L * D, generator=g).to(dev)
    dy = torch.randn(N, L * D, generator=g).to(dev)

    def run():
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y = layer(xg, ei)
        y.backward(dy)
        m = layer.multi_head_attention
        return [t.detach().cpu().numpy() for t in (y, xg.grad, m.in_proj_weight.grad, m.out_proj.weight.grad)]

    fast = run()
    assert graph_cache.get(ei, N).hub_dst_chunks > 0 and graph_cache.get(ei, N).hub_src_chunks > 0
    monkeypatch.setenv('AMPCONV_FORCE_GENERIC', '1')
    slow = run()
    for name, a, b in zip(['y', 'dx', 'gWin', 'gWo'], fast, slow):
        assert_close_scaled(a, b, name)


def test_message_aggregate_gradients_match_fused(dev):
    """The decomposed public path message() -> aggregate() (PyG's propagate by hand) must train like the
    fused propagate(): same output, same gradients w.r.t. x and the parameters (aggregate()
    used to return a tensor without grad_fn)."""
    from ampnet_amd import AMPConv
    torch.manual_seed(3)
    N, E, L, D, H = 40, 300, 20, 128, 4
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    x = torch.randn(N, L * D, device=dev)
    dy = torch.randn(N, L * D, device=dev)
    ei = torch.randint(0, N, (2, E), device=dev)
    ei[1, ei[1] == 3] = 4                                      # node 3 receives nothing

    xa = x.clone().requires_grad_(True)
    ya = layer(xa, ei)
    ya.backward(dy)
    ga = [xa.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    layer.zero_grad(set_to_none=True)

    xb = x.clone().requires_grad_(True)
    msg = layer.message(xb.index_select(0, ei[1]), xb.index_select(0, ei[0]))
    yb = layer.aggregate(msg, ei[1], dim_size=N)
    assert yb.requires_grad
    yb.backward(dy)
    gb = [xb.grad] + [p.grad for p in layer.parameters()]
    assert_close_scaled(yb.detach().cpu().numpy(), ya.detach().cpu().numpy(), 'y (message+aggregate vs fused)')
    assert (yb[3] == 0).all()
    for name, a, b in zip(['dx', 'g_in_w', 'g_in_b', 'g_out_w', 'g_out_b'], ga, gb):
        assert_close_scaled(b.cpu().numpy(), a.cpu().numpy(), name + ' (message+aggregate vs fused)')


def test_message_aggregate_gradients_scaled_projections(dev, monkeypatch):
    """The same with the fp32 projections' scaled two-plane mode forced on (the pre-gathered x_i / x_j path measures two
    input maxima and records the K | V projection's)."""
    from ampnet_amd.conv import functional as F_
    monkeypatch.setattr(F_, 'PROJ_SCALED_MIN_ELEMENTS', 0)
    test_message_aggregate_gradients_match_fused(dev)
    test_message_matches_per_edge_reference(dev)


@pytest.mark.parametrize('shape', [(600, 9000, 20, 256, 8), (500, 6000, 20, 64, 4), (400, 5000, 13, 64, 2), (300, 3000, 40, 100, 2),
                                   (300, 2500, 2, 128, 8), (200, 2000, 5, 12, 4)],
                         ids=['t4_dh32', 't4_dh16', 'mfma_L13', 'block_L40_dh50', 'small_L2', 'generic_dh3'])
def test_backward_passes_record_the_maximum_of_what_they_write(shape, dev, monkeypatch):
    """`out_absmax` of ampconv_bwd_edge_dst / _src (include/ampconv.h): every kernel family, with long segments (the
    combine pass writes those rows), equals max |dQKV| exactly; the scale source of the fp32 projections' scaled mode."""
    from ampnet_amd import EdgeCSR, _lib
    from ampnet_amd.conv import functional as F_
    monkeypatch.setattr(_lib, 'HUB_CHUNK', 64)
    N, E, L, D, H = shape
    dh = D // H
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(E)
    ei = torch.randint(0, N, (2, E), device=dev, generator=g)
    ei[1, : E // 8] = 3                  # long segments on both sides
    ei[0, E // 8: E // 4] = 5
    csr = EdgeCSR(ei, N)
    qkv = torch.randn(N * L, 3 * D, device=dev, generator=g)
    dobar = torch.randn(N * L, D, device=dev, generator=g) * 1e-3
    dqkv = torch.full((N * L, 3 * D), float('nan'), device=dev)
    Qv, Kv, Vv = (F_._view(qkv, i * D, L, dh) for i in range(3))
    dQv, dKv, dVv = (F_._view(dqkv, i * D, L, dh) for i in range(3))
    dOv = F_._view(dobar, 0, L, dh)
    nstat = lib.ampconv_softmax_stats_bytes(E, L, D, H, _lib.AMPCONV_F32)
    stats = torch.empty(max(nstat, 4) // 4, device=dev) if nstat else None
    spos = csr.csc_positions() if nstat else None
    am = torch.zeros(2, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan, nch, ws = csr.hub_args('dst', L, D, 1)
    assert nch > 0
    _lib.check(lib.ampconv_bwd_edge_dst(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L, D, H, dQv, plan,
                                        nch, F_._ptr(ws), F_._ptr(spos), F_._ptr(stats), am[0:1].data_ptr(),
                                        _lib.AMPCONV_F32, st), 'dst')
    plan, nch, ws = csr.hub_args('src', L, D, 2)
    _lib.check(lib.ampconv_bwd_edge_src(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(), csr.cinv.data_ptr(),
                                        N, L, D, H, dKv, dVv, plan, nch, F_._ptr(ws), F_._ptr(stats), am[1:2].data_ptr(),
                                        _lib.AMPCONV_F32, st), 'src')
    assert torch.isfinite(dqkv).all()
    assert float(am[0]) == float(dqkv[:, :D].abs().max()) > 0
    assert float(am[1]) == float(dqkv[:, D:].abs().max()) > 0
    # bf16 storage takes no maximum
    hq = qkv.bfloat16()
    hv = [F_._view(hq, i * D, L, dh) for i in range(3)]
    rc = lib.ampconv_bwd_edge_dst(*hv, F_._view(dobar.bfloat16(), 0, L, dh), csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L,
                                  D, H, F_._view(dqkv.bfloat16(), 0, L, dh), None, 0, None, None, None, am[0:1].data_ptr(),
                                  _lib.AMPCONV_BF16, st)
    assert rc == -2


def test_retain_attention_auto(dev, monkeypatch):
    """Default 'auto': small graphs keep what the lazy per-edge attributes need, large projection buffers are
    dropped and reading the attributes says so instead of pinning them (61 GB at config 4)."""
    from ampnet_amd import AMPConv
    torch.manual_seed(4)
    layer = AMPConv(128, 4).to(dev)
    assert layer.retain_attention == 'auto'
    x = torch.randn(50, 20 * 128, device=dev)
    ei = torch.randint(0, 50, (2, 200), device=dev)
    with torch.no_grad():
        layer(x, ei)
    assert layer.attn_output_weights.shape == (200, 20, 20)
    assert layer.attn_output.shape == (200, 20, 128)
    monkeypatch.setenv('AMPCONV_RETAIN_LIMIT_MB', '0')
    with torch.no_grad():
        layer(x, ei)
    assert layer._attn_ctx is None
    with pytest.raises(RuntimeError, match='not retained'):
        layer.attn_output_weights
    layer.retain_attention = True
    with torch.no_grad():
        layer(x, ei)
    assert layer.attn_output_weights is not None
    with torch.no_grad():                                     # parameters changed since forward: warn
        layer.multi_head_attention.out_proj.bias.add_(1.0)
    with pytest.warns(RuntimeWarning, match='out_proj'):
        layer.attn_output


def test_gemm_precision_is_restored(dev):
    """A plain fp32 GEMM after a 'bf16x3' block is bit-equal to the same GEMM before it (the switches are
    process-global; functional.gemm_precision restores them under a lock)."""
    from ampnet_amd.conv.functional import gemm_precision
    torch.manual_seed(6)
    a = torch.randn(4096, 512, device=dev)
    b = torch.randn(512, 768, device=dev)
    before = a @ b
    with gemm_precision('bf16x3'):
        inside = a @ b
    after = a @ b
    assert torch.equal(before, after)
    assert inside.shape == before.shape


def test_bench_launches_two_ranks(dev):
    """`python bench.py --gpus 2` starts its two ranks itself and reports n_gpus = 2 (gloo here: both ranks
    share the one card of the test box; on the 8-GPU node the same path runs over RCCL)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, AMPCONV_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'tiny',
                        '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-alt-gemm'], env=env, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['value'] > 0
    # one line diagnoses a scaling run: per-rank kernel and all-reduce times, and BASELINE config 4 as written
    # (GraphSAINT batches on the same ranks) beside the full-graph number
    assert [r['rank'] for r in out['per_rank_ms']] == [0, 1] and all(r['bwd_edge_src'] > 0 and r['allreduce'] > 0
                                                                   for r in out['per_rank_ms'])
    assert out['allreduce_ms'] > 0
    saint = out['saint']
    assert saint['steps'] >= 20 and saint['value'] > 0 and saint['nodes_avg'] > 0 and saint['edges_avg'] > 0
    assert saint['sampler_ms'] > 0 and saint['allreduce_ms'] > 0 and len(saint['per_rank_ms']) == 2
    # the line itself shows that the collective library saw both ranks
    assert out['dist'] == {'backend': 'gloo', 'world_size': 2, 'ranks_seen': 2}


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_rmat_graph_with_id_structured_degrees(dtype, dev):
    """BASELINE config 5's generator at test size (R-MAT, no label permutation: a node's degree follows the bits of its
    id, long segments on both sides, isolated nodes, a row count that is NOT a power of two): the main launch walks the
    rows in the scrambled order of common.h, the long segments go through the chunk plan -- against the fp64 oracle."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    from ampnet_amd import AMPConv
    from oracle.ampconv_numpy import AMPConvOracle
    N, E, L, D, H = 3000, 36000, 20, 64, 2
    g = torch.Generator().manual_seed(77)
    ei = bench.rmat_edges(12, E, g, 'cpu')                      # ids in [0, 4096) ...
    ei = ei[:, (ei < N).all(dim=0)]                             # ... cut to N = 3000 rows
    E = ei.size(1)
    deg_in, deg_out = torch.bincount(ei[1], minlength=N), torch.bincount(ei[0], minlength=N)
    assert deg_in.max() > 64 and deg_out.max() > 64 and (deg_in == 0).sum() > 50      # hubs and empty rows exist
    torch.manual_seed(78)
    layer = AMPConv(D, H)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    layer = layer.to(dev).to(tdt)
    x = torch.randn(N, L * D, generator=g).to(tdt)
    dy = torch.randn(N, L * D, generator=g).to(tdt)
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg, ei.to(dev))
    y.backward(dy.to(dev))
    m = layer.multi_head_attention
    f64 = lambda t: t.detach().float().cpu().numpy().astype(np.float64)
    o = AMPConvOracle(f64(m.in_proj_weight), f64(m.in_proj_bias), f64(m.out_proj.weight), f64(m.out_proj.bias),
                      H, dtype=np.float64, edge_chunk=2048)
    y_ref, _ = o.forward(f64(x), ei.numpy(), need_weights=False)
    dx_ref, dWin, dbin, dWo, dbo = o.backward(f64(dy))
    tol = dict(atol=2e-2, rtol=2e-2) if dtype == 'bf16' else {}
    assert_close_scaled(f64(y), y_ref, 'y', **tol)
    assert (f64(y)[deg_in.numpy() == 0] == 0).all()
    assert_close_scaled(f64(xg.grad), dx_ref, 'dx', **tol)
    assert_close_scaled(f64(m.in_proj_weight.grad), dWin, 'g_in_proj_weight', **tol)
    assert_close_scaled(f64(m.in_proj_bias.grad), dbin, 'g_in_proj_bias', **tol)
    assert_close_scaled(f64(m.out_proj.weight.grad), dWo, 'g_out_proj_weight', **tol)
    assert_close_scaled(f64(m.out_proj.bias.grad), dbo, 'g_out_proj_bias', **tol)


def test_bench_rccl_path_single_rank(dev):
    """The N-rank run's RCCL code path (nccl process group with device_id, parameter broadcast, gradient all-reduce,
    barriers, MAX reduction of the time) executed by ONE rank on the one GPU of the test box -- two ranks cannot share
    a card under RCCL, so this is as much of the `--gpus N` path as a single GPU can run."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, AMPCONV_BENCH_FORCE_DIST='1', AMPCONV_DIST_BACKEND='nccl', HSA_ENABLE_IPC_MODE_LEGACY='0',
               MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):      # bench.py picks a free port itself
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'tiny', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--no-alt-gemm'], env=env, timeout=600, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['value'] > 0
    assert out['allreduce_ms'] > 0 and out['saint']['allreduce_ms'] > 0         # both RCCL all-reduces ran
    assert out['edge_phase_hbm']['frac_of_8TBps'] > 0 and 'traffic_source' in out['roofline']
    assert out['dist'] == {'backend': 'nccl', 'world_size': 1, 'ranks_seen': 1}
    # one rank: the side measurements of the default N = 1 line run as well (a Cora-sized stand-in here)
    assert out['extra_workloads']['cora']['value'] > 0


def test_bench_default_line_carries_saint_and_extras_without_a_process_group(dev):
    """the plain N = 1 run (no process group) measures the GraphSAINT-batch mode and the extra
    workloads too; stdout stays ONE JSON line."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'AMPCONV_BENCH_FORCE_DIST'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'tiny', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--no-alt-gemm'], env=env, timeout=600, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert 'dist' not in out and out['n_gpus'] == 1
    assert out['saint']['steps'] >= 20 and out['saint']['value'] > 0 and out['saint']['allreduce_ms'] == 0
    assert out['extra_workloads']['cora']['ms_per_step'] > 0


def test_nt4_kernels_match_round1_tiling(dev, tmp_path):
    """The 4x4x1 tail kernels (default) against the same build with AMPCONV_{FWD,DST,SRC}_NT4=0 (every product on
    16x16x4, the round-1 tiling): an independent MFMA formulation of the same arithmetic.  The switches are read once
    per process, so the round-1 tiling runs in a child process."""
    import subprocess
    import sys
    from conftest import ROOT
    script = f'''
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from ampnet_amd import AMPConv
torch.manual_seed(12)
dev = torch.device("cuda:0")
N, E, L, D, H = 500, 7000, 20, 256, 8
layer = AMPConv(D, H).to(dev)
with torch.no_grad():
    layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
    layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
g = torch.Generator(device=dev).manual_seed(13)
x = torch.randn(N, L * D, device=dev, generator=g).requires_grad_(True)
dy = torch.randn(N, L * D, device=dev, generator=g)
ei = torch.randint(0, N, (2, E), device=dev, generator=g)
ei[1, :300] = 3
ei[0, 300:500] = 4
y = layer(x, ei)
y.backward(dy)
m = layer.multi_head_attention
np.savez(sys.argv[1], y=y.detach().cpu().numpy(), dx=x.grad.cpu().numpy(), gw=m.in_proj_weight.grad.cpu().numpy(),
         gb=m.in_proj_bias.grad.cpu().numpy(), gow=m.out_proj.weight.grad.cpu().numpy())
'''
    outs = {}
    for name, extra in (('nt4', {}), ('r1', {'AMPCONV_FWD_NT4': '0', 'AMPCONV_DST_NT4': '0', 'AMPCONV_SRC_NT4': '0'})):
        path = str(tmp_path / f'{name}.npz')
        r = subprocess.run([sys.executable, '-c', script, path], env=dict(os.environ, **extra), timeout=600,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = dict(np.load(path))
    for k in outs['nt4']:
        assert_close_scaled(outs['nt4'][k], outs['r1'][k], k + ' (4x4x1 tail vs 16x16x4 tiling)')


def test_config2_literal_two_layers_cora_size(dev):
    """BASELINE config 2 at its literal size: two stacked AMPConv layers with ReLU between (amp_gcn.py:248-262,
    cora_benchmark_graphsaint.py:159-163: the final full-graph evaluation) on N = 2 708 nodes / E = 10 556 edges,
    L = 20, D = 128, H = 4, against the reference-shaped CPU restatement (oracle/ampconv_torch.py: gather -> stock
    nn.MultiheadAttention -> scatter-mean, ~2 GB, a few seconds): outputs, input gradient and all eight parameter
    gradients at the fp32 tolerance.  Cora itself is not available offline: Cora-shaped synthetic
    graph, 5 278 undirected pairs in both directions, power-law degrees clipped to Cora's maximum 168."""
    from ampnet_amd import AMPConv
    from oracle.ampconv_torch import RefShapedAMPConv
    N, L, D, H, pairs = 2708, 20, 128, 4, 5278
    g = torch.Generator().manual_seed(7)
    w = torch.arange(1, N + 1, dtype=torch.float64).pow(-0.8)              # power-law endpoint weights
    a = torch.multinomial(w, pairs, replacement=True, generator=g)
    b = torch.randint(0, N, (pairs,), generator=g)
    ei = torch.cat([torch.stack([a, b]), torch.stack([b, a])], dim=1)
    deg = torch.bincount(ei[1], minlength=N)
    assert ei.size(1) == 10556 and int(deg.max()) > 40 and int((deg == 0).sum()) > 0
    keep = torch.ones(ei.size(1), dtype=torch.bool)
    for n in torch.nonzero(deg > 168).flatten().tolist():                  # clip to Cora's maximum degree
        idx = torch.nonzero(ei[1] == n).flatten()[168:]
        ei[1, idx] = (n + 1 + torch.arange(idx.numel())) % N
    x = torch.randn(N, L * D, generator=g)
    dy = torch.randn(N, L * D, generator=g)
    torch.manual_seed(3)
    refs = [RefShapedAMPConv(D, H), RefShapedAMPConv(D, H)]
    for r in refs:
        with torch.no_grad():
            r.multi_head_attention.in_proj_bias.normal_(0, 0.1)
            r.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    xr = x.clone().requires_grad_(True)
    yr = refs[1](torch.relu(refs[0](xr, ei)), ei)
    yr.backward(dy)
    layers = []
    for r in refs:
        l = AMPConv(D, H).to(dev)
        l.load_state_dict(r.state_dict())                                   # same keys as the reference
        layers.append(l)
    xg = x.to(dev).requires_grad_(True)
    eid = ei.to(dev)
    y = layers[1](torch.relu(layers[0](xg, eid)), eid)
    y.backward(dy.to(dev))
    assert_close_scaled(y.detach().cpu().numpy(), yr.detach().numpy(), 'y')
    assert (y[deg.to(dev) == 0] == 0).all()
    assert_close_scaled(xg.grad.cpu().numpy(), xr.grad.numpy(), 'dx')
    for i, (l, r) in enumerate(zip(layers, refs)):
        for (name, p), q in zip(l.named_parameters(), r.parameters()):
            assert_close_scaled(p.grad.cpu().numpy(), q.grad.numpy(), f'layer {i + 1} {name}.grad')


@pytest.mark.parametrize('gemm', ['native', 'fp32'])
def test_key_bias_gradient_choice_is_inside_tolerance_by_construction(gemm, dev):
    """The key bias shifts every score of a softmax row equally, so its gradient is EXACTLY 0 in exact arithmetic;
    autograd returns rounding noise (the reference's own value in the fixtures is below 1e-6 of the largest entry of
    in_proj_bias.grad).  The two projection modes treat it differently -- 'native' returns the true column sums of
    dK (noise of the same size), the library-GEMM path writes exact zeros and takes the value-bias third from the
    identity colsum(dV) = colsum_masked(dY) Wo (functional.py) -- and both must sit inside the fixture tolerance
    because the reference's entries are that small, not by luck of one fixture."""
    for name in ('cora_L20_D128_H4', 'cfg4_L20_D256_H8', 'cfg3_L20_D128_H8'):
        g = load_golden(os.path.join(GOLDEN_DIR, name + '.npz'))
        D = int(g['D'])
        ref = g['g_in_proj_bias']
        scale = float(np.abs(ref).max())
        assert float(np.abs(ref[D:2 * D]).max()) <= 1e-6 * scale          # the reference's own key-bias gradient
        layer = _layer(g, dev, gemm=gemm)
        x = torch.from_numpy(g['x']).to(dev).requires_grad_(True)
        y = layer(x, torch.from_numpy(g['edge_index']).to(dev))
        (y * torch.from_numpy(g['dy']).to(dev)).sum().backward()
        gb = layer.multi_head_attention.in_proj_bias.grad.cpu().numpy()
        assert float(np.abs(gb[D:2 * D]).max()) <= 1e-5 * scale
        if gemm == 'fp32':
            assert (gb[D:2 * D] == 0).all()
        assert_close_scaled(gb[2 * D:], ref[2 * D:], 'value-bias gradient')
        assert_close_scaled(gb[:D], ref[:D], 'query-bias gradient')


@pytest.mark.parametrize('shape', [(1, 128, 8), (2, 64, 8), (3, 128, 4), (4, 128, 8), (4, 256, 8), (2, 32, 2), (1, 256, 8)],
                         ids=lambda s: 'L%d_D%d_H%d' % s)
def test_short_sequences_small_kernels(shape, dev, monkeypatch):
    """L <= 4 runs on the one-wave-per-row VALU kernels of csrc/edge_small.hip (the L = 1 / L = 4 sweeps of
    config 3; the reference's XOR harness runs L = 2).  Checked against the fp32 oracle at the flat tolerance, and
    against the tile kernels (AMPCONV_SMALL=0) -- with a 300-in-edge and a 200-out-edge hub, so that the long-segment
    passes (main + chunk + combine) of the new family run too."""
    from ampnet_amd import AMPConv, graph_cache
    from oracle.ampconv_numpy import AMPConvOracle
    L, D, H = shape
    N, E = 260, 2200
    g = torch.Generator().manual_seed(100 * L + D + H)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :300] = 5
    ei[0, 300:500] = 9
    ei[1, ei[1] == 17] = 18                                  # node 17 receives nothing -> exact zero row
    torch.manual_seed(L + D)
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.2)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.2)
    x = torch.randn(N, L * D, generator=g)
    dy = torch.randn(N, L * D, generator=g)

    def run():
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xg = x.to(dev).requires_grad_(True)
        y = layer(xg, ei.to(dev))
        y.backward(dy.to(dev))
        m = layer.multi_head_attention
        return [t.detach().cpu().numpy() for t in (y, xg.grad, m.in_proj_weight.grad, m.in_proj_bias.grad,
                                                    m.out_proj.weight.grad, m.out_proj.bias.grad)]

    small = run()
    m = layer.multi_head_attention
    o = AMPConvOracle(m.in_proj_weight.detach().cpu().numpy(), m.in_proj_bias.detach().cpu().numpy(),
                      m.out_proj.weight.detach().cpu().numpy(), m.out_proj.bias.detach().cpu().numpy(), H)
    y_ref, _ = o.forward(x.numpy(), ei.numpy(), need_weights=False)
    refs = (y_ref,) + tuple(o.backward(dy.numpy()))
    names = ['y', 'dx', 'g_in_proj_weight', 'g_in_proj_bias', 'g_out_proj_weight', 'g_out_proj_bias']
    for name, a, b in zip(names, small, refs):
        # dx of the 200-out-edge hub is a sum over its edges: an accumulated quantity like the parameter gradients
        assert_close_scaled(a, b, f'{name} (small kernels vs oracle, L={L} D={D} H={H})', scaled=True if name == 'dx' else None)
    assert (small[0][17] == 0).all()
    monkeypatch.setenv('AMPCONV_SMALL', '0')
    tiles = run()
    for name, a, b in zip(names, small, tiles):
        assert_close_scaled(a, b, f'{name} (small vs tile kernels)', scaled=True if name == 'dx' else None)


@pytest.mark.parametrize('path', [p for p in SINGLE if 'cora' in os.path.basename(p) or 'cfg4' in os.path.basename(p)][:2],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_graphed_layer_matches_eager(path, dev):
    """ampnet_amd.GraphedAMPConv: forward and backward of a layer on a fixed graph recorded as two HIP graphs and
    replayed -- the reference's golden vectors through the captured path (same tolerance), bit-identical to the eager
    path on fresh inputs, and replays follow new inputs and new parameter values."""
    from ampnet_amd import GraphedAMPConv
    g = load_golden(path)
    layer = _layer(g, dev)
    x = torch.from_numpy(g['x']).to(dev).requires_grad_(True)
    ei = torch.from_numpy(g['edge_index']).to(dev)
    dy = torch.from_numpy(g['dy']).to(dev)
    fast = GraphedAMPConv(layer, x, ei)
    layer.zero_grad(set_to_none=True)
    y = fast(x)
    (y * dy).sum().backward()
    assert_close_scaled(y.detach().cpu().numpy(), g['y'], 'y')
    assert_close_scaled(x.grad.cpu().numpy(), g['dx'], 'dx')
    gw, gb, gow, gob = _grads(layer)
    assert_close_scaled(gw, g['g_in_proj_weight'], 'g_in_proj_weight')
    assert_close_scaled(gb, g['g_in_proj_bias'], 'g_in_proj_bias')
    assert_close_scaled(gow, g['g_out_proj_weight'], 'g_out_proj_weight')
    assert_close_scaled(gob, g['g_out_proj_bias'], 'g_out_proj_bias')
    # fresh inputs and changed parameters: replay == eager, bit for bit
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.add_(0.05)
    x2 = (torch.randn_like(x) * 0.7).requires_grad_(True)
    dy2 = torch.randn_like(dy)
    layer.zero_grad(set_to_none=True)
    y2 = fast(x2)
    y2.backward(dy2)
    got = (y2.detach().clone(), x2.grad.clone()) + tuple(torch.from_numpy(a) for a in _grads(layer))
    x3 = x2.detach().clone().requires_grad_(True)
    layer.zero_grad(set_to_none=True)
    y3 = layer(x3, ei)
    y3.backward(dy2)
    want = (y3.detach(), x3.grad) + tuple(torch.from_numpy(a) for a in _grads(layer))
    for a, b, name in zip(got, want, ('y', 'dx', 'gw', 'gb', 'gow', 'gob')):
        assert torch.equal(a.cpu(), b.cpu()), name
