"""GPU parity of the plane-format edge phase (csrc/edge_mfma_f16x2.hip; include/ampconv.h "edge phase on fp16 PLANES"):
Q | K | V and dObar as two fp16 planes of the power-of-two-scaled value, every product three v_mfma_f32_16x16x32_f16
partial products.  What it replaces in the reference is what the fp32 edge kernels replace (amp_conv.py:39 -> torch
functional.py:6578-6594 and its autograd backward), so the checker is the same fp64 oracle at the same FLAT fp32
tolerance; the golden vectors run through this path in tests/test_gpu_parity.py
(test_golden_single_layer_scaled_projections: every fixture with dh = 32 or 16 and embed_dim % 128 == 0)."""
import numpy as np
import pytest
import torch

from conftest import assert_close_scaled

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from ampnet_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


class _Calls:
    """Counts the calls of C entry points (the ctypes function objects are wrapped, arguments pass through)."""

    def __init__(self, monkeypatch, names):
        from ampnet_amd import _lib
        lib = _lib.load()
        self.n = {k: 0 for k in names}
        for k in names:
            real = getattr(lib, k)

            def wrapped(*a, _real=real, _k=k):
                self.n[_k] += 1
                return _real(*a)
            monkeypatch.setattr(lib, k, wrapped)


PLANE_CALLS = ('ampconv_fwd_edge_planes', 'ampconv_bwd_edge_dst_planes', 'ampconv_bwd_edge_src_planes')
F32_CALLS = ('ampconv_fwd_edge', 'ampconv_bwd_edge_dst', 'ampconv_bwd_edge_src')


def _make(N, E, L, D, H, dev, seed=3, x_scale=1.0, hub=True):
    from ampnet_amd import AMPConv
    torch.manual_seed(seed)
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(N, L * D, generator=g) * x_scale
    dy = torch.randn(N, L * D, generator=g)
    ei = torch.randint(0, N - 20, (2, E), generator=g)            # last 20 nodes isolated
    if hub:
        ei[1, : E // 10] = 3                                      # long CSR segment
        ei[0, E // 10: E // 5] = 7                                # long CSC segment
    return layer, x, dy, ei


def _oracle(layer, x, dy, ei, H):
    from oracle.ampconv_numpy import AMPConvOracle
    m = layer.multi_head_attention
    o = AMPConvOracle(m.in_proj_weight.detach().cpu().numpy(), m.in_proj_bias.detach().cpu().numpy(),
                      m.out_proj.weight.detach().cpu().numpy(), m.out_proj.bias.detach().cpu().numpy(),
                      H, dtype=np.float64, edge_chunk=2048)
    y_ref, _ = o.forward(x.numpy(), ei.numpy(), need_weights=False)
    return (y_ref,) + tuple(o.backward(dy.numpy()))


def _run(layer, x, dy, ei, dev):
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg, ei.to(dev))
    y.backward(dy.to(dev))
    m = layer.multi_head_attention
    out = (y.detach().cpu().numpy(), xg.grad.cpu().numpy(), m.in_proj_weight.grad.cpu().numpy(),
           m.in_proj_bias.grad.cpu().numpy(), m.out_proj.weight.grad.cpu().numpy(), m.out_proj.bias.grad.cpu().numpy())
    for p in m.parameters():
        p.grad = None
    return out


NAMES = ('y', 'dx', 'g_in_proj_weight', 'g_in_proj_bias', 'g_out_proj_weight', 'g_out_proj_bias')


@pytest.mark.parametrize('shape', [(1500, 15000, 20, 256, 8), (900, 9000, 20, 128, 4), (700, 6000, 13, 128, 4),
                                   (600, 5000, 17, 256, 8), (400, 3000, 5, 128, 4), (300, 2500, 1, 128, 4),
                                   (1500, 9000, 20, 128, 8), (600, 5000, 17, 256, 16), (400, 3000, 3, 128, 8)],
                         ids=['cfg4_like', 'cora_like', 'L13', 'L17', 'L5', 'L1', 'cfg3_like_dh16', 'L17_dh16', 'L3_dh16'])
@pytest.mark.parametrize('stats', [True, False], ids=['stats', 'own_softmax'])
def test_plane_path_vs_fp64_oracle(shape, stats, dev, monkeypatch):
    """The whole layer through the plane-format edge passes (forced onto small graphs), with long segments in both
    directions and isolated nodes, against the fp64 oracle at the flat fp32 tolerance; the three plane entry points
    are what ran."""
    from ampnet_amd.conv import functional as F_
    monkeypatch.setattr(F_, 'PROJ_SCALED_MIN_ELEMENTS', 0)
    monkeypatch.setattr(F_, 'PLANES_MIN_L', 1)               # (by default L <= 4 stays on the short-sequence kernels)
    monkeypatch.setattr(F_, 'SOFTMAX_STATS', stats)          # hand-off of (log-sum-exp, delta) from the dst to the src pass
    N, E, L, D, H = shape
    layer, x, dy, ei = _make(N, E, L, D, H, dev)
    calls = _Calls(monkeypatch, PLANE_CALLS + F32_CALLS)
    got = _run(layer, x, dy, ei, dev)
    assert all(calls.n[k] == 1 for k in PLANE_CALLS) and all(calls.n[k] == 0 for k in F32_CALLS), calls.n
    want = _oracle(layer, x, dy, ei, H)
    for name, a, b in zip(NAMES, got, want):
        assert_close_scaled(a, b, name)
    assert (got[0][N - 20:] == 0).all(), 'rows with no in-edge must be exactly 0'


def test_plane_path_error_table_vs_fp64(dev, monkeypatch):
    """max error / max entry against fp64 at a BASELINE-shaped size, next to the fp32-MFMA edge kernels on the same
    inputs: the plane path must stay under the bar the fp32 path is held to (5e-6; DESIGN.md 4c has the table)."""
    from ampnet_amd.conv import functional as F_
    monkeypatch.setattr(F_, 'PROJ_SCALED_MIN_ELEMENTS', 0)
    N, E, L, D, H = 1500, 15000, 20, 256, 8
    layer, x, dy, ei = _make(N, E, L, D, H, dev, seed=31, x_scale=2.0, hub=False)
    want = _oracle(layer, x, dy, ei, H)
    rows = {}
    for planes in (True, False):
        monkeypatch.setattr(F_, 'EDGE_PLANES', planes)
        got = _run(layer, x, dy, ei, dev)
        rows[planes] = [float(np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(got, want)]
    for i, name in enumerate(NAMES):
        print(f'[err] {name}: planes {rows[True][i]:.2e}   fp32 edge kernels {rows[False][i]:.2e}')
    assert max(rows[True][:2]) < 5e-6, rows[True]


def test_plane_path_large_scores(dev, monkeypatch):
    """Saturated softmaxes (scores of several hundred): the scores meet their scale only inside the exponential, the
    masks are -inf; against the oracle (fp64) and the fp32 edge kernels."""
    from ampnet_amd.conv import functional as F_
    monkeypatch.setattr(F_, 'PROJ_SCALED_MIN_ELEMENTS', 0)
    N, E, L, D, H = 300, 3000, 20, 128, 4
    layer, x, dy, ei = _make(N, E, L, D, H, dev, seed=5, x_scale=12.0)
    want = _oracle(layer, x, dy, ei, H)
    got = _run(layer, x, dy, ei, dev)
    monkeypatch.setattr(F_, 'EDGE_PLANES', False)
    ref32 = _run(layer, x, dy, ei, dev)
    assert np.isfinite(got[0]).all() and np.isfinite(got[1]).all()
    for name, a, b, c in zip(NAMES, got, want, ref32):
        # (saturated rows: dx is a difference of large terms; the bar is the one the fp32 kernels' own large-score test
        # uses, and the fp32 kernels' error on the same inputs is printed beside it)
        assert_close_scaled(a, b, name, atol=3e-5, rtol=3e-4, scaled=True)
        print(f'[err] {name}: planes {np.abs(a - b).max():.3e}  fp32 kernels {np.abs(c - b).max():.3e}  max |ref| {np.abs(b).max():.3e}')


def test_plane_path_is_bitwise_reproducible_and_keeps_side_outputs(dev, monkeypatch):
    """Two runs give identical bits (no atomics in any sum); attn_output_weights / attn_output are served from the
    plane-format projection buffer (read back through ampconv_planes_to_f32) and match the oracle."""
    from ampnet_amd.conv import functional as F_
    from oracle.ampconv_numpy import AMPConvOracle
    monkeypatch.setattr(F_, 'PROJ_SCALED_MIN_ELEMENTS', 0)
    N, E, L, D, H = 500, 4000, 20, 128, 4
    layer, x, dy, ei = _make(N, E, L, D, H, dev, seed=9)
    a = _run(layer, x, dy, ei, dev)
    w = layer.attn_output_weights.cpu().numpy()
    b = _run(layer, x, dy, ei, dev)
    for name, u, v in zip(NAMES, a, b):
        assert np.array_equal(u, v), name
    m = layer.multi_head_attention
    o = AMPConvOracle(m.in_proj_weight.detach().cpu().numpy(), m.in_proj_bias.detach().cpu().numpy(),
                      m.out_proj.weight.detach().cpu().numpy(), m.out_proj.bias.detach().cpu().numpy(), H, dtype=np.float64)
    _, w_ref = o.forward(x.numpy(), ei.numpy())
    assert_close_scaled(w, w_ref, 'attn_output_weights')
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-5)


def test_range_guard_sends_wide_operands_to_the_exact_kernels(dev, monkeypatch):
    """One scale per tensor serves rows within 2^12 of the tensor's maximum.  A node 2^24 times the rest (x), or a
    gradient row 2^24 times the rest (dY), must not cost the other rows their accuracy: ampconv_absmax_stats sees it,
    the call takes the exact kernels (six-product projections, fp32 edge passes), and the OTHER rows meet the flat
    tolerance against fp64."""
    from ampnet_amd.conv import functional as F_
    monkeypatch.setattr(F_, 'PROJ_SCALED_MIN_ELEMENTS', 0)
    N, E, L, D, H = 400, 3000, 20, 128, 4
    layer, x, dy, ei = _make(N, E, L, D, H, dev, seed=13, hub=False)
    ei = ei[:, (ei[0] != 5) & (ei[1] != 5)]                       # node 5 talks to nobody: its rows stay its own
    calls = _Calls(monkeypatch, PLANE_CALLS + F32_CALLS)
    # (a) outlier in x
    xo = x.clone()
    xo[5] *= float(1 << 24)
    got = _run(layer, xo, dy, ei, dev)
    assert all(calls.n[k] == 0 for k in PLANE_CALLS) and all(calls.n[k] == 1 for k in F32_CALLS), calls.n
    want = _oracle(layer, xo, dy, ei, H)
    keep = np.arange(N) != 5
    assert_close_scaled(got[0][keep], want[0][keep], 'y')
    assert_close_scaled(got[1][keep], want[1][keep], 'dx')
    # (b) forward on planes, outlier in dY: the backward pass falls back, reading the plane-format projections back
    for k in calls.n:
        calls.n[k] = 0
    dyo = dy.clone()
    dyo[9] *= float(1 << 24)
    got = _run(layer, x, dyo, ei, dev)
    assert calls.n['ampconv_fwd_edge_planes'] == 1 and calls.n['ampconv_bwd_edge_dst_planes'] == 0, calls.n
    assert calls.n['ampconv_bwd_edge_dst'] == 1 and calls.n['ampconv_bwd_edge_src'] == 1, calls.n
    want = _oracle(layer, x, dyo, ei, H)
    touched = np.zeros(N, dtype=bool)                             # rows that see node 9's gradient: 9 and its sources
    touched[9] = True
    touched[ei[0][ei[1] == 9].numpy()] = True
    assert_close_scaled(got[0], want[0], 'y')
    assert_close_scaled(got[1][~touched], want[1][~touched], 'dx')


def test_absmax_stats_and_the_stats_cache(dev):
    """ampconv_absmax_stats against numpy (maximum; smallest non-zero maximum of a 32-channel group; NaN / inf skipped,
    all-zero groups not counted), and operand_stats' cache: same object + same version -> no second pass."""
    from ampnet_amd.conv import functional as F_
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1000, 256, generator=g)
    x[17, 64:96] *= 1e-5                       # one head slot far below the rest
    x[40, 0:32] = 0.0                          # an all-zero slot does not count
    x[3, 5] = float('nan')
    x[4, 7] = float('inf')
    xd = x.to(dev)
    st, narrow = F_.operand_stats(xd)
    fin = np.where(np.isfinite(x.numpy()), np.abs(x.numpy()), 0.0)
    seg = fin.reshape(1000, 8, 32).max(-1)
    want = (fin.max(), seg[seg > 0].min())
    got = st.cpu().numpy()
    assert got[0] == np.float32(want[0]) and got[1] == np.float32(want[1]), (got, want)
    assert not narrow
    y = torch.randn(64, 128, generator=g).to(dev)
    s1, n1 = F_.operand_stats(y, key=y)
    s2, n2 = F_.operand_stats(y, key=y)
    assert s1 is s2 and n1 and n2
    y.mul_(2.0)                                # version bump: measured again
    s3, _ = F_.operand_stats(y, key=y)
    assert s3 is not s1 and float(s3[0]) == 2.0 * float(s1[0])


@pytest.mark.parametrize('dh', [32, 16])
def test_plane_projection_round_trip(dh, dev):
    """ampconv_proj_rows_planes -> ampconv_planes_to_f32 equals the fp32-output scaled product to 2^-20 of the bound
    (two 11-bit planes), with bias, with the in-degree division (row_scale) and exact zero rows for empty segments;
    the recorded maximum covers the requested columns only."""
    from ampnet_amd.conv import functional as F_
    g = torch.Generator().manual_seed(4)
    M, K, N, L = 4000, 256, 768, 20
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    am = F_.absmax(a)
    img = F_.proj_image(w)
    bound = torch.empty(1, device=dev)
    F_.proj_out_bound(w, False, b, am, bound)
    ref = F_.proj_rows(a, img, b, amax=am)
    assert float(bound) >= float(ref.abs().max())
    rec = torch.zeros(1, device=dev)
    pl = F_.proj_rows_planes(a, img, bound, b, amax=am, out_amax=rec, amax_col0=512, dh=dh)
    back = F_.planes_to_f32(pl, bound, dh)
    err = float((back - ref).abs().max())
    print(f'[err] planes round trip: {err:.3e}, bound {float(bound):.3e}, max |out| {float(ref.abs().max()):.3e}')
    assert err <= float(bound) * 2.0 ** -20
    assert float(rec) == float(ref[:, 512:].abs().max())
    # rows divided by the in-degree of their node; rows of nodes without in-edges exactly zero
    nn = M // L
    deg = torch.randint(0, 6, (nn,), generator=g)
    rowptr = torch.zeros(nn + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
    pl = F_.proj_rows_planes(a, img, bound, None, rowptr=rowptr.to(dev), L=L, row_scale=1, amax=am, dh=dh)
    back = F_.planes_to_f32(pl, bound, dh).cpu()
    ref0 = F_.proj_rows(a, img, None, amax=am).cpu()
    inv = torch.where(deg > 0, 1.0 / deg.clamp(min=1).float(), torch.zeros(nn)).repeat_interleave(L)[:, None]
    assert float((back - ref0 * inv).abs().max()) <= float(bound) * 2.0 ** -20
    assert (back[(inv == 0).expand_as(back)] == 0).all()
