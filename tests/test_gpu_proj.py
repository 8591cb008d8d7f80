"""Node-phase projection kernels (csrc/proj_gemm.hip, csrc/proj_gemm_bf16.hip) through the C ABI against fp64 products.

What they replace: torch functional.py:5785-5862 (`_in_projection_packed`), :6600 (out-projection) and
their autograd backward.  Tolerance: the kernels are fp32-grade by construction (operands split exactly
into three bf16 terms, six partial products, fp32 accumulation), so the bar is the error of torch's own
fp32 GEMM against the same fp64 product -- at most 2x it (plus one ulp of slack) -- not a loose absolute."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _err(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max())


def _rowptr(n_nodes, empty, dev):
    deg = torch.ones(n_nodes, dtype=torch.int32)
    deg[list(empty)] = 0
    rp = torch.zeros(n_nodes + 1, dtype=torch.int32)
    rp[1:] = torch.cumsum(deg, 0)
    return rp.to(dev), deg.bool()


@pytest.mark.parametrize('M,K,N', [(5, 128, 128), (129, 128, 384), (1000, 256, 768), (4096, 768, 256),
                                   (20 * 333, 256, 256), (777, 384, 128),
                                   # ragged shapes (tiles padded inside): the reference's default embed_dim = 100
                                   # (amp_gcn.py:21-35), 64 (fixture wide_L4_D64_H8), 8 and 4 (toy fixtures)
                                   (40 * 77, 100, 300), (333, 300, 100), (501, 100, 100), (260, 64, 192), (37, 8, 24),
                                   (5, 4, 12), (700, 200, 100)])
@pytest.mark.parametrize('transpose', [False, True])
def test_proj_rows_matches_fp64(M, K, N, transpose):
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + K + N)
    a = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(*((K, N) if transpose else (N, K)), device=dev, generator=g) * 0.1
    bias = torch.randn(N, device=dev, generator=g)
    Wnk = W.t() if transpose else W
    ref = a.double() @ Wnk.double().t() + bias.double()
    out = F_.proj_rows(a, F_.proj_image(W, transpose=transpose), bias)
    lib_err = _err(torch.addmm(bias, a, Wnk.t()), ref)
    assert _err(out, ref) <= 2 * lib_err + 1e-7, (_err(out, ref), lib_err)
    # no bias, bitwise reproducible
    out2 = F_.proj_rows(a, F_.proj_image(W, transpose=transpose))
    assert torch.equal(out2, F_.proj_rows(a, F_.proj_image(W, transpose=transpose)))
    assert _err(out2, ref - bias.double()) <= 2 * lib_err + 1e-7


def test_proj_rows_strided_input_and_sliced_weight():
    # the non-shared path: w_in[:D] / w_in[D:] slices, inputs that are column blocks of a wider buffer
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(3)
    D, M = 128, 300
    w_in = torch.randn(3 * D, D, device=dev, generator=g) * 0.1
    wide = torch.randn(M, 3 * D, device=dev, generator=g)
    out = F_.proj_rows(wide[:, D:], F_.proj_image(w_in[D:], transpose=True))      # [M, 2D] @ [2D, D]
    ref = wide[:, D:].double() @ w_in[D:].double()
    assert _err(out, ref) < 1e-6


@pytest.mark.parametrize('L,empty', [(20, (0, 7, 49)), (1, (3,)), (7, ())])
def test_proj_rows_mask_gives_exact_zero_rows(L, empty):
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    n_nodes, K, N = 50, 128, 256
    rp, has = _rowptr(n_nodes, empty, dev)
    g = torch.Generator(device=dev).manual_seed(L)
    a = torch.randn(n_nodes * L, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.1
    bias = torch.randn(N, device=dev, generator=g)
    out = F_.proj_rows(a, F_.proj_image(W), bias, rp, L).view(n_nodes, L * N)
    ref = (a.double() @ W.double().t() + bias.double()).view(n_nodes, L * N) * has.to(dev)[:, None]
    assert _err(out, ref) < 1e-6
    for n in empty:
        assert (out[n] == 0).all()


@pytest.mark.parametrize('M,Na,Nb', [(37, 128, 128), (20 * 271, 384, 128), (5000, 768, 256), (70001, 256, 256),
                                     (40 * 77, 300, 100), (4000, 100, 100), (999, 192, 64), (61, 24, 8), (20 * 40, 200, 100),
                                     # several stages per row slice (the rows of two stages are in flight), with the mask
                                     (20 * 3000, 128, 128), (20 * 9000, 256, 256), (20 * 5000, 384, 256)])
@pytest.mark.parametrize('masked', [False, True])
def test_proj_wgrad_matches_fp64(M, Na, Nb, masked):
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + Na)
    a = torch.randn(M, Na, device=dev, generator=g)
    b = torch.randn(M, Nb, device=dev, generator=g)
    L = 20 if M % 20 == 0 else 1
    rp = flag = None
    if masked:
        rp, has = _rowptr(M // L, (0, 5, M // L - 1), dev)
        flag = has.to(dev).repeat_interleave(L)[:, None].double()
    am = a.double() * flag if masked else a.double()
    ref_dw, ref_cs = am.t() @ b.double(), am.sum(0)
    dw = torch.empty(Na, Nb, device=dev)
    cs = torch.empty(Na, device=dev)
    F_.proj_wgrad(a, b, dw, cs, rp, L)
    lib_err = _err((a * flag.float() if masked else a).t() @ b, ref_dw)
    assert _err(dw, ref_dw) <= 2 * lib_err + 1e-7, (_err(dw, ref_dw), lib_err)
    assert float((cs.double() - ref_cs).abs().max()) <= 1e-6 * float(am.abs().sum(0).max()) + 1e-6
    dw2 = torch.empty_like(dw)
    cs2 = torch.empty_like(cs)
    F_.proj_wgrad(a, b, dw2, cs2, rp, L)
    assert torch.equal(dw, dw2) and torch.equal(cs, cs2)            # fixed slices, ordered sum


@pytest.mark.parametrize('Na,Nb', [(128, 128), (384, 128), (256, 256)])
@pytest.mark.parametrize('scaled', [False, True])
def test_proj_wgrad_masked_is_reproducible(Na, Nb, scaled):
    """300 launches of the masked weight-gradient product on the same inputs give the same bits (builds of round 4 with
    two stages of rows in flight and the compiler's own interleaving of transposed fragment reads and MFMAs were wrong in
    7-100 % of the launches at 128 x 128: DESIGN.md 4a; the kernel now fences the reads)."""
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(Na + Nb)
    M, L = 60000, 20
    a = torch.randn(M, Na, device=dev, generator=g)
    b = torch.randn(M, Nb, device=dev, generator=g)
    deg = (torch.rand(M // L, device=dev, generator=g) < 0.8).int()
    rp = torch.zeros(M // L + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    amax = (F_.absmax(a), F_.absmax(b)) if scaled else None
    flag = deg.bool().repeat_interleave(L)[:, None]
    ref = (a.double() * flag).t() @ b.double()
    dw0, cs0 = torch.empty(Na, Nb, device=dev), torch.empty(Na, device=dev)
    F_.proj_wgrad(a, b, dw0, cs0, rp, L, amax=amax)
    assert _err(dw0, ref) < 2e-6
    bad = 0
    for it in range(300):
        dw, cs = torch.empty(Na, Nb, device=dev), torch.empty(Na, device=dev)
        if it % 2:
            torch.randn(1 << 18, device=dev, generator=g)          # something else on the stream in between
        F_.proj_wgrad(a, b, dw, cs, rp, L, amax=amax)
        bad += int(not (torch.equal(dw, dw0) and torch.equal(cs, cs0)))
    assert bad == 0, f'{bad} of 300 launches differ'


@pytest.mark.parametrize('K,N', [(256, 768), (768, 256), (100, 300), (128, 128)])
@pytest.mark.parametrize('scaled', [False, True])
def test_proj_rows_is_reproducible(K, N, scaled):
    """The row kernel streams its rows with inline-assembly loads and hand-counted waits (csrc/proj_gemm.hip): 200
    launches on the same inputs, other work on the stream in between, give the same bits -- full tiles, ragged K / N
    (embed_dim 100), with and without the mask, both plane modes."""
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(K + N)
    L, n_nodes = 20, 2500
    M = n_nodes * L
    a = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.1
    bias = torch.randn(N, device=dev, generator=g)
    deg = (torch.rand(n_nodes, device=dev, generator=g) < 0.8).int()
    rp = torch.zeros(n_nodes + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    img = F_.proj_image(W)
    amax = F_.absmax(a) if scaled else None
    ref = a.double() @ W.double().t() + bias.double()
    for mask in (None, rp):
        out0 = F_.proj_rows(a, img, bias, mask, L, amax=amax)
        want = ref if mask is None else ref * deg.bool().repeat_interleave(L)[:, None]
        assert _err(out0, want) < 2e-6
        bad = 0
        for it in range(200):
            if it % 2:
                torch.randn(1 << 18, device=dev, generator=g)
            bad += int(not torch.equal(F_.proj_rows(a, img, bias, mask, L, amax=amax), out0))
        assert bad == 0, f'{bad} of 200 launches differ (mask: {mask is not None})'


def test_proj_wgrad_into_row_block_views():
    # the non-shared path writes d in_proj_weight in two row blocks of one tensor
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(11)
    D, M = 128, 999
    dq, dkv = torch.randn(M, D, device=dev, generator=g), torch.randn(M, 2 * D, device=dev, generator=g)
    x = torch.randn(M, D, device=dev, generator=g)
    dw, db = torch.empty(3 * D, D, device=dev), torch.empty(3 * D, device=dev)
    F_.proj_wgrad(dq, x, dw[:D], db[:D])
    F_.proj_wgrad(dkv, x, dw[D:], db[D:])
    ref = torch.cat([dq, dkv], 1).double().t() @ x.double()
    assert _err(dw, ref) < 1e-6
    assert _err(db, torch.cat([dq, dkv], 1).double().sum(0)) < 1e-6


def test_unsupported_shapes_are_refused_not_miscomputed():
    from ampnet_amd import _lib
    lib = _lib.load()
    F32, BF16 = _lib.AMPCONV_F32, _lib.AMPCONV_BF16
    assert lib.ampconv_proj_supported(256, 256, F32) == 1 and lib.ampconv_proj_supported(768, 256, F32) == 1
    assert lib.ampconv_proj_supported(100, 100, F32) == 1 and lib.ampconv_proj_supported(300, 100, F32) == 1   # padded inside
    assert lib.ampconv_proj_supported(3, 3, F32) == 0 and lib.ampconv_proj_supported(102, 100, F32) == 0       # not float4 rows
    assert lib.ampconv_proj_supported(256, 256, BF16) == 1 and lib.ampconv_proj_supported(24, 72, BF16) == 1
    assert lib.ampconv_proj_supported(100, 100, BF16) == 0      # bf16 rows move in pieces of 8 elements: library GEMM
    assert lib.ampconv_proj_supported(256, 256, 7) == 0
    assert lib.ampconv_proj_rows(None, 100, 10, 100, None, 100, None, None, 0, None, 100, None, 0, None, None, F32, None) == -1  # null pointers
    assert lib.ampconv_proj_rows(None, 3, 10, 3, None, 3, None, None, 0, None, 3, None, 0, None, None, F32, None) == -1
    assert lib.ampconv_proj_rows(None, 256, 10, 256, None, 256, None, None, 0, None, 256, None, 0, None, None, BF16, None) == -1
    assert lib.ampconv_proj_rows(None, 256, 10, 256, None, 256, None, None, 0, None, 256, None, 0, None, None, 7, None) == -2   # dtype


# ---- fp32 storage, scaled two-plane mode (include/ampconv.h "SCALED MODE"): same bar as the six-product kernels
@pytest.mark.parametrize('M,K,N', [(5, 128, 128), (1000, 256, 768), (4096, 768, 256), (20 * 333, 256, 256),
                                   (777, 384, 128), (260, 100, 300), (333, 100, 100), (37, 4, 12)])
@pytest.mark.parametrize('magnitude', [1.0, 3e-7, 2e4])
def test_proj_rows_scaled_matches_fp64(M, K, N, magnitude):
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + K + N)
    a = torch.randn(M, K, device=dev, generator=g) * magnitude
    W = torch.randn(N, K, device=dev, generator=g) * 0.1
    bias = torch.randn(N, device=dev, generator=g) * magnitude
    img = F_.proj_image(W)
    ref = a.double() @ W.double().t() + bias.double()
    am = F_.absmax(a)
    assert float(am) == float(a.abs().max())
    om = torch.zeros(1, device=dev)
    out = F_.proj_rows(a, img, bias, amax=am, out_amax=om)
    assert _err(out, ref) < 1e-6
    assert float(om) == float(out.abs().max())                      # the recorded maximum of what was written
    assert torch.equal(out, F_.proj_rows(a, img, bias, amax=am))      # bitwise reproducible
    # an upper bound of the maximum within a few binades serves as well
    assert _err(F_.proj_rows(a, img, bias, amax=am * 50), ref) < 1e-6
    # six-product form on the same inputs (twice the accumulator roundings: 1.07e-6 at K = 768 on one of these seeds)
    assert _err(F_.proj_rows(a, img, bias), ref) < 2e-6


def test_proj_rows_scaled_wide_range_rows_and_nonfinite():
    """Rows 2^-16 of the tensor's maximum keep fp32-level RELATIVE accuracy; NaN / infinity stay in their rows."""
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(7)
    M, K, N = 512, 256, 256
    a = torch.randn(M, K, device=dev, generator=g)
    a *= torch.exp2(-torch.arange(M, device=dev) % 17).float()[:, None]       # row scales 1 ... 2^-16
    W = torch.randn(N, K, device=dev, generator=g) * 0.1
    img = F_.proj_image(W)
    ref = a.double() @ W.double().t()
    out = F_.proj_rows(a, img, amax=F_.absmax(a))
    rel = (out.double() - ref).abs().amax(1) / ref.abs().amax(1)
    assert float(rel.max()) < 2e-6, float(rel.max())
    a[3, 5] = float('nan')
    a[9, 0] = float('inf')
    am = F_.absmax(a)
    assert torch.isfinite(am).all()                                   # skipped by the maximum
    out = F_.proj_rows(a, img, amax=am)
    assert torch.isnan(out[3]).all() and not torch.isfinite(out[9]).any()
    keep = torch.ones(M, dtype=torch.bool, device=dev)
    keep[3] = keep[9] = False
    assert torch.isfinite(out[keep]).all()
    assert float(((out.double() - ref).abs().amax(1) / ref.abs().amax(1))[keep].max()) < 2e-6


@pytest.mark.parametrize('M,Na,Nb', [(37, 128, 128), (20 * 271, 384, 128), (5000, 768, 256), (70001, 256, 256),
                                     (999, 100, 100), (20 * 40, 300, 100), (20 * 3000, 128, 128), (20 * 9000, 256, 256),
                                     (20 * 5000, 384, 256)])
@pytest.mark.parametrize('masked', [False, True])
def test_proj_wgrad_scaled_matches_fp64(M, Na, Nb, masked):
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + Na)
    a = torch.randn(M, Na, device=dev, generator=g) * 1e-4           # gradients are small
    b = torch.randn(M, Nb, device=dev, generator=g) * 3.0
    L = 20 if M % 20 == 0 else 1
    rp = flag = None
    if masked:
        rp, has = _rowptr(M // L, (0, 5, M // L - 1), dev)
        flag = has.to(dev).repeat_interleave(L)[:, None]
    am = a.double() * flag.double() if masked else a.double()
    ref_dw, ref_cs = am.t() @ b.double(), am.sum(0)
    dw, cs = torch.empty(Na, Nb, device=dev), torch.empty(Na, device=dev)
    amax = (F_.absmax(a), F_.absmax(b))
    F_.proj_wgrad(a, b, dw, cs, rp, L, amax=amax)
    lib_err = _err((a * flag.float() if masked else a).t() @ b, ref_dw)
    assert _err(dw, ref_dw) <= 2 * lib_err + 1e-7, (_err(dw, ref_dw), lib_err)
    assert float((cs.double() - ref_cs).abs().max()) <= 1e-6 * float(am.abs().sum(0).max()) + 1e-12
    dw2, cs2 = torch.empty_like(dw), torch.empty_like(cs)
    F_.proj_wgrad(a, b, dw2, cs2, rp, L, amax=amax)
    assert torch.equal(dw, dw2) and torch.equal(cs, cs2)
    F_.proj_wgrad(a, b, dw2, cs2, rp, L)                              # six-product form: the same bar
    assert _err(dw2, ref_dw) <= 2 * lib_err + 1e-7


def test_absmax_strided_bf16_and_merge():
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(1)
    wide = torch.randn(3000, 768, device=dev, generator=g)
    wide[17, 300] = -77.5
    wide[5, 700] = 1e9                                                # outside the column block below
    assert float(F_.absmax(wide[:, 256:512])) == 77.5
    m = F_.absmax(wide[:, :256])
    F_.absmax(wide[:, 512:], out=m)                                   # merged into an existing maximum
    assert float(m) == 1e9
    h = wide[:, :256].bfloat16()
    assert float(F_.absmax(h)) == float(h.float().abs().max())
    assert float(F_.absmax(torch.zeros(8, 4, device=dev))) == 0.0


# ---- bf16 storage (csrc/proj_gemm_bf16.hip; BASELINE config 5).  One bf16 x bf16 product per element pair is exact in
# fp32, the sum is fp32, the result is rounded to bf16 ONCE: the bar is half a bf16 ulp (2^-8 relative) of the fp64
# product of the same bf16 inputs, plus the fp32 accumulation noise.
def _bf16_close(out, ref, what):
    err = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() * 1.001 + 2e-5 * float(ref.abs().max()) + 1e-30
    bad = err > tol
    assert not bad.any(), f'{what}: {int(bad.sum())}/{bad.numel()} beyond half a bf16 ulp, max err {float(err.max()):.3e}'


@pytest.mark.parametrize('M,K,N', [(5, 128, 128), (129, 128, 384), (1000, 256, 768), (4096, 768, 256),
                                   (20 * 333, 256, 256), (777, 384, 128), (128 * 17, 256, 768),
                                   # ragged: K not a multiple of 64, N not of 128 (tiles padded inside)
                                   (260, 64, 192), (37, 8, 24), (333, 72, 40), (700, 200, 104), (501, 24, 72)])
@pytest.mark.parametrize('transpose', [False, True])
def test_proj_rows_bf16_matches_fp64(M, K, N, transpose):
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + K + N)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(*((K, N) if transpose else (N, K)), device=dev, generator=g) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev, generator=g).bfloat16()
    Wnk = W.t() if transpose else W
    ref = a.double() @ Wnk.double().t() + bias.double()
    out = F_.proj_rows(a, F_.proj_image(W, transpose=transpose), bias)
    assert out.dtype == torch.bfloat16
    _bf16_close(out, ref, 'rows + bias')
    out2 = F_.proj_rows(a, F_.proj_image(W, transpose=transpose))
    assert torch.equal(out2, F_.proj_rows(a, F_.proj_image(W, transpose=transpose)))      # bitwise reproducible
    _bf16_close(out2, ref - bias.double(), 'rows')


def test_proj_rows_bf16_strided_input_and_sliced_weight():
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(3)
    D, M = 128, 300
    w_in = (torch.randn(3 * D, D, device=dev, generator=g) * 0.1).bfloat16()
    wide = torch.randn(M, 3 * D, device=dev, generator=g).bfloat16()
    out = F_.proj_rows(wide[:, D:], F_.proj_image(w_in[D:], transpose=True))      # [M, 2D] @ [2D, D]
    _bf16_close(out, wide[:, D:].double() @ w_in[D:].double(), 'strided rows')
    # a row block that starts at an odd element (not 16-byte aligned): copied, not refused
    out = F_.proj_rows(wide[:, 1:1 + D], F_.proj_image(w_in[:D], transpose=True))
    _bf16_close(out, wide[:, 1:1 + D].double() @ w_in[:D].double(), 'unaligned rows')


@pytest.mark.parametrize('L,empty', [(20, (0, 7, 49)), (1, (3,)), (7, ())])
def test_proj_rows_bf16_mask_gives_exact_zero_rows(L, empty):
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    n_nodes, K, N = 50, 128, 256
    rp, has = _rowptr(n_nodes, empty, dev)
    g = torch.Generator(device=dev).manual_seed(L)
    a = torch.randn(n_nodes * L, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev, generator=g).bfloat16()
    out = F_.proj_rows(a, F_.proj_image(W), bias, rp, L).view(n_nodes, L * N)
    ref = (a.double() @ W.double().t() + bias.double()).view(n_nodes, L * N) * has.to(dev)[:, None]
    _bf16_close(out, ref, 'masked rows')
    for n in empty:
        assert (out[n] == 0).all()


@pytest.mark.parametrize('M,Na,Nb', [(37, 128, 128), (20 * 271, 384, 128), (5000, 768, 256), (70001, 256, 256),
                                     (999, 192, 64), (61, 24, 8), (20 * 40, 200, 104), (31, 256, 256), (20 * 1000, 72, 24)])
@pytest.mark.parametrize('masked', [False, True])
def test_proj_wgrad_bf16_matches_fp64(M, Na, Nb, masked):
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + Na)
    a = torch.randn(M, Na, device=dev, generator=g).bfloat16()
    b = torch.randn(M, Nb, device=dev, generator=g).bfloat16()
    L = 20 if M % 20 == 0 else 1
    rp = flag = None
    if masked:
        rp, has = _rowptr(M // L, (0, 5, M // L - 1), dev)
        flag = has.to(dev).repeat_interleave(L)[:, None]
        b = b * flag.to(b.dtype)        # bf16: the PRODUCT is unmasked; the layer's other operand (Obar) is 0 there
    am = a.double() * flag.double() if masked else a.double()
    ref_dw, ref_cs = am.t() @ b.double(), am.sum(0)
    dw = torch.empty(Na, Nb, device=dev, dtype=torch.bfloat16)
    cs = torch.empty(Na, device=dev, dtype=torch.bfloat16)
    F_.proj_wgrad(a, b, dw, cs, rp, L)
    # fp32 partial sums of exact products, ordered fp32 sum over the slices, one rounding: half an ulp + the fp32 noise of a
    # sum of M terms of size ~1
    noise = 4e-7 * M ** 0.5
    err = (dw.double() - ref_dw).abs()
    assert not (err > 2.0 ** -8 * ref_dw.abs() * 1.001 + noise).any(), float(err.max())
    err = (cs.double() - ref_cs).abs()
    assert not (err > 2.0 ** -8 * ref_cs.abs() * 1.001 + noise).any(), float(err.max())
    dw2, cs2 = torch.empty_like(dw), torch.empty_like(cs)
    F_.proj_wgrad(a, b, dw2, cs2, rp, L)
    assert torch.equal(dw, dw2) and torch.equal(cs, cs2)            # fixed slices, ordered sum


def test_proj_wgrad_bf16_masked_rows_do_not_leak_into_the_column_sums():
    """NaN in a row whose node has no in-edge must not reach the (masked) column sums."""
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(2)
    L, n_nodes, D = 20, 40, 256
    rp, has = _rowptr(n_nodes, (3, 11), dev)
    a = torch.randn(n_nodes * L, D, device=dev, generator=g).bfloat16()
    b = torch.randn(n_nodes * L, D, device=dev, generator=g).bfloat16()
    a[3 * L + 2, 17] = float('nan')
    a[11 * L:12 * L] = float('inf')
    b[3 * L:4 * L] = 0
    b[11 * L:12 * L] = 0
    dw = torch.empty(D, D, device=dev, dtype=torch.bfloat16)
    cs = torch.empty(D, device=dev, dtype=torch.bfloat16)
    F_.proj_wgrad(a, b, dw, cs, rp, L)
    assert torch.isfinite(cs).all()
    flag = has.to(dev).repeat_interleave(L)[:, None]
    ref = torch.where(flag, a.double(), torch.zeros((), dtype=torch.float64, device=dev)).sum(0)
    assert float((cs.double() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-4


@pytest.mark.parametrize('K,N', [(256, 768), (768, 256), (72, 40)])
def test_proj_rows_bf16_is_reproducible(K, N):
    """The all-DMA bf16 row kernel (hand-counted waits throughout): 200 launches, same bits -- plain, masked, node list."""
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(K + N)
    L, n_nodes = 20, 2500
    a = torch.randn(n_nodes * L, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev, generator=g).bfloat16()
    deg = (torch.rand(n_nodes, device=dev, generator=g) < 0.6).int()
    rp = torch.zeros(n_nodes + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    ids = torch.cat([deg.nonzero().flatten().int(), torch.zeros(8, dtype=torch.int32, device=dev)])
    cnt = int(deg.sum())
    img = F_.proj_image(W)
    for kw in (dict(), dict(rowptr=rp), dict(nodes=(ids, cnt))):
        out0 = torch.zeros(n_nodes * L, N, device=dev, dtype=torch.bfloat16)
        F_.proj_rows(a, img, bias, L=L, out=out0, **kw)
        bad = 0
        for it in range(200):
            out = torch.zeros_like(out0)
            if it % 2:
                torch.randn(1 << 18, device=dev, generator=g)
            F_.proj_rows(a, img, bias, L=L, out=out, **kw)
            bad += int(not torch.equal(out, out0))
        assert bad == 0, f'{bad} of 200 launches differ ({list(kw)})'


@pytest.mark.parametrize('Na,Nb', [(128, 128), (768, 256)])
def test_proj_wgrad_bf16_is_reproducible(Na, Nb):
    """The bf16 weight-gradient product reads its operands with the same transposing LDS instruction as the fp32 one
    (whose compiler-chosen schedule was racy: DESIGN.md 4a) and carries the same fence: 200 launches, same bits."""
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(Na)
    M, L = 60000, 20
    a = torch.randn(M, Na, device=dev, generator=g).bfloat16()
    b = torch.randn(M, Nb, device=dev, generator=g).bfloat16()
    deg = (torch.rand(M // L, device=dev, generator=g) < 0.8).int()
    rp = torch.zeros(M // L + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    for mask in (rp, None):
        dw0 = torch.empty(Na, Nb, device=dev, dtype=torch.bfloat16)
        cs0 = torch.empty(Na, device=dev, dtype=torch.bfloat16)
        F_.proj_wgrad(a, b, dw0, cs0, mask, L)
        bad = 0
        for it in range(200):
            dw, cs = torch.empty_like(dw0), torch.empty_like(cs0)
            if it % 2:
                torch.randn(1 << 18, device=dev, generator=g)
            F_.proj_wgrad(a, b, dw, cs, mask, L)
            bad += int(not (torch.equal(dw, dw0) and torch.equal(cs, cs0)))
        assert bad == 0, f'{bad} of 200 launches differ'


def test_proj_wgrad_bf16_into_row_block_views():
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(11)
    D, M = 128, 999
    dq = torch.randn(M, D, device=dev, generator=g).bfloat16()
    dkv = torch.randn(M, 2 * D, device=dev, generator=g).bfloat16()
    x = torch.randn(M, D, device=dev, generator=g).bfloat16()
    dw = torch.empty(3 * D, D, device=dev, dtype=torch.bfloat16)
    db = torch.empty(3 * D, device=dev, dtype=torch.bfloat16)
    F_.proj_wgrad(dq, x, dw[:D], db[:D])
    F_.proj_wgrad(dkv, x, dw[D:], db[D:])
    cat = torch.cat([dq, dkv], 1).double()
    ref = cat.t() @ x.double()
    assert float((dw.double() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-4
    assert float((db.double() - cat.sum(0)).abs().max()) <= 2.0 ** -8 * float(cat.sum(0).abs().max()) + 1e-4


def _sparse_graph(N, E, n_silent, dev, seed):
    """E random edges among the first N - n_silent nodes (the last n_silent have no edge at all), plus a few nodes that
    only send and a few that only receive."""
    g = torch.Generator(device=dev).manual_seed(seed)
    act = N - n_silent
    src = torch.randint(0, act // 2, (E,), device=dev, generator=g)           # senders: the first half
    dst = torch.randint(act // 4, act, (E,), device=dev, generator=g)         # receivers: the last three quarters
    return torch.stack([src, dst])


@pytest.mark.parametrize('N,E,n_silent', [(97, 300, 30), (1000, 2500, 400), (40, 20, 0)])
def test_active_nodes_lists(N, E, n_silent):
    from ampnet_amd import EdgeCSR
    from ampnet_amd import graph as G_
    dev = _dev()
    ei = _sparse_graph(N, E, n_silent, dev, N)
    csr = EdgeCSR(ei, N)
    old = G_.ACTIVE_LIST_FRACTION
    G_.ACTIVE_LIST_FRACTION = 1.0
    try:
        lists = csr.active_nodes()
    finally:
        G_.ACTIVE_LIST_FRACTION = old
    has_in = torch.zeros(N, dtype=torch.bool, device=dev)
    has_in[ei[1]] = True
    has_out = torch.zeros(N, dtype=torch.bool, device=dev)
    has_out[ei[0]] = True
    for name, has in (('in', has_in), ('out', has_out), ('any', has_in | has_out)):
        ids, cnt, ptr = lists[name]
        ref = has.nonzero().flatten().int()
        assert cnt == ref.numel() and ids.numel() == cnt + 8
        assert torch.equal(ids[:cnt], ref)
        assert torch.equal(ptr[1:] - ptr[:-1], has.int()) and int(ptr[0]) == 0


@pytest.mark.parametrize('L', [16, 20, 33, 64, 128])
@pytest.mark.parametrize('K,N', [(256, 768), (768, 256), (256, 256), (72, 40)])
def test_proj_rows_bf16_node_list(L, K, N):
    """Listed nodes: bitwise the rows of the full product; every other row of `out` untouched."""
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    n_nodes = 61
    g = torch.Generator(device=dev).manual_seed(L + K)
    keep = torch.rand(n_nodes, device=dev, generator=g) < 0.55
    keep[0], keep[-1] = True, False
    ids = torch.cat([keep.nonzero().flatten().int(), torch.zeros(8, dtype=torch.int32, device=dev)])
    cnt = int(keep.sum())
    a = torch.randn(n_nodes * L, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev, generator=g).bfloat16()
    img = F_.proj_image(W)
    full = F_.proj_rows(a, img, bias)
    out = torch.full((n_nodes * L, N), 7.0, dtype=torch.bfloat16, device=dev)
    F_.proj_rows(a, img, bias, L=L, nodes=(ids, cnt), out=out)
    rows = keep.repeat_interleave(L)
    assert torch.equal(out[rows], full[rows])
    assert (out[~rows] == 7.0).all()
    # a list of every node = the plain call; an empty list writes nothing
    every = torch.arange(n_nodes + 8, dtype=torch.int32, device=dev)
    assert torch.equal(F_.proj_rows(a, img, bias, L=L, nodes=(every, n_nodes)), full)
    out.fill_(7.0)
    F_.proj_rows(a, img, bias, L=L, nodes=(ids, 0), out=out)
    assert (out == 7.0).all()
    # into a column block of a wider matrix (the layer's Q | K | V thirds)
    wide = torch.full((n_nodes * L, N + 128), 7.0, dtype=torch.bfloat16, device=dev)
    F_.proj_rows(a, img, bias, L=L, nodes=(ids, cnt), out=wide[:, 128:])
    assert torch.equal(wide[:, 128:][rows], full[rows]) and (wide[:, :128] == 7.0).all()


@pytest.mark.parametrize('L,n_nodes', [(16, 50), (20, 700), (20, 4001), (48, 333), (128, 70)])
@pytest.mark.parametrize('Na,Nb', [(256, 256), (768, 256), (72, 24)])
def test_proj_wgrad_bf16_node_list(L, n_nodes, Na, Nb):
    """Sum over the listed nodes' rows only: NaN in every other row must not matter; fp64 reference over those rows."""
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(L + n_nodes + Na)
    keep = torch.rand(n_nodes, device=dev, generator=g) < 0.5
    keep[1], keep[-1] = True, True
    keep[0] = False
    ids = torch.cat([keep.nonzero().flatten().int(), torch.zeros(8, dtype=torch.int32, device=dev)])
    cnt = int(keep.sum())
    rows = keep.repeat_interleave(L)
    a = torch.randn(n_nodes * L, Na, device=dev, generator=g).bfloat16()
    b = torch.randn(n_nodes * L, Nb, device=dev, generator=g).bfloat16()
    a[~rows] = float('nan')
    b[~rows] = float('nan')
    dw = torch.empty(Na, Nb, device=dev, dtype=torch.bfloat16)
    cs = torch.empty(Na, device=dev, dtype=torch.bfloat16)
    F_.proj_wgrad(a, b, dw, cs, L=L, nodes=(ids, cnt))
    ad, bd = a[rows].double(), b[rows].double()
    ref_dw, ref_cs = ad.t() @ bd, ad.sum(0)
    noise = 4e-7 * (cnt * L) ** 0.5
    err = (dw.double() - ref_dw).abs()
    assert not (err > 2.0 ** -8 * ref_dw.abs() * 1.001 + noise).any(), float(err.max())      # (NaN compares False: checked next)
    assert torch.isfinite(dw).all() and torch.isfinite(cs).all()
    err = (cs.double() - ref_cs).abs()
    assert not (err > 2.0 ** -8 * ref_cs.abs() * 1.001 + noise).any(), float(err.max())
    dw2, cs2 = torch.empty_like(dw), torch.empty_like(cs)
    F_.proj_wgrad(a, b, dw2, cs2, L=L, nodes=(ids, cnt))
    assert torch.equal(dw, dw2) and torch.equal(cs, cs2)
    # the compacted rows through the plain kernel: the same sum (slice boundaries differ: not bitwise)
    dw3, cs3 = torch.empty_like(dw), torch.empty_like(cs)
    F_.proj_wgrad(a[rows].contiguous(), b[rows].contiguous(), dw3, cs3)
    assert float((dw3.double() - dw.double()).abs().max()) <= 2.0 ** -7 * float(ref_dw.abs().max()) + noise


def test_proj_node_list_argument_checks():
    from ampnet_amd import _lib
    lib = _lib.load()
    dev = _dev()
    F32, BF16 = _lib.AMPCONV_F32, _lib.AMPCONV_BF16
    x = torch.zeros(64 * 20, 256, device=dev, dtype=torch.bfloat16)
    ids = torch.zeros(72, dtype=torch.int32, device=dev)
    from ampnet_amd.conv import functional as F_
    img = F_.proj_image(torch.zeros(256, 256, device=dev, dtype=torch.bfloat16))[0]
    rp = torch.zeros(65, dtype=torch.int32, device=dev)

    def rows(L, rowptr, dtype, n):
        return lib.ampconv_proj_rows(x.data_ptr(), 256, x.size(0), 256, img.data_ptr(), 256, None,
                                     None if rowptr is None else rowptr.data_ptr(), L, x.data_ptr(), 256,
                                     ids.data_ptr(), n, None, None, dtype, None)
    assert rows(20, None, F32, 4) == -2          # fp32 storage takes no list
    assert rows(20, rp, BF16, 4) == -1           # list and mask together
    assert rows(8, None, BF16, 4) == -1          # L < 16
    assert rows(200, None, BF16, 4) == -1        # L > 128
    assert rows(20, None, BF16, 65) == -1        # more listed nodes than the matrix has


def test_bf16_layer_node_lists_match_the_full_projections():
    """The layer on a graph where 40 % of the nodes have no edge: with node lists (default) and without
    (AMPCONV_NODE_LISTS=0 semantics) the outputs and every gradient agree -- bitwise on the rows, to the slice order on
    the weight gradients."""
    from ampnet_amd import AMPConv
    from ampnet_amd.conv import functional as F_
    dev = _dev()
    torch.manual_seed(4)
    N, E, L, D, H = 500, 3000, 20, 256, 8
    ei = _sparse_graph(N, E, 200, dev, 1)
    layer = AMPConv(D, H).to(dev).to(torch.bfloat16)
    x0 = torch.randn(N, L * D, device=dev).bfloat16()
    dy = torch.randn(N, L * D, device=dev).bfloat16()
    res = []
    for on in (True, False):
        old = F_.NODE_LISTS
        F_.NODE_LISTS = on
        try:
            x = x0.clone().requires_grad_(True)
            layer.zero_grad()
            y = layer(x, ei)
            y.backward(dy)
            m = layer.multi_head_attention
            res.append([y.detach(), x.grad] + [p.grad.clone() for p in (m.in_proj_weight, m.in_proj_bias,
                                                                          m.out_proj.weight, m.out_proj.bias)])
        finally:
            F_.NODE_LISTS = old
    from ampnet_amd import graph_cache
    assert graph_cache.get(ei, N).active_nodes() is not None          # the list path really ran
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2:], res[1][2:]):
        assert torch.isfinite(a).all()
        assert float((a.double() - b.double()).abs().max()) <= 2.0 ** -7 * float(b.double().abs().max()) + 1e-6


def test_bf16_layer_runs_no_library_gemm(monkeypatch):
    """BASELINE config 5's storage mode end to end in HIP: with the default 'native' projections a
    bf16 layer issues no torch GEMM, reduction or cast of its own -- torch.addmm / mm / bmm are made to raise."""
    from ampnet_amd import AMPConv
    dev = _dev()
    torch.manual_seed(9)
    N, E, L, D, H = 300, 2400, 20, 256, 8
    layer = AMPConv(D, H).to(dev).to(torch.bfloat16)
    x = torch.randn(N, L * D, device=dev).bfloat16().requires_grad_(True)
    dy = torch.randn(N, L * D, device=dev).bfloat16()
    ei = torch.randint(0, N, (2, E), device=dev)

    def boom(*a, **k):
        raise AssertionError('library GEMM called in native bf16 mode')
    for name in ('addmm', 'mm', 'bmm', 'matmul'):
        monkeypatch.setattr(torch, name, boom)
    monkeypatch.setattr(torch.Tensor, 'mm', boom)
    monkeypatch.setattr(torch.Tensor, 'matmul', boom)
    y = layer(x, ei)
    y.backward(dy)
    m = layer.multi_head_attention
    for t in (y, x.grad, m.in_proj_weight.grad, m.in_proj_bias.grad, m.out_proj.weight.grad, m.out_proj.bias.grad):
        assert t.dtype == torch.bfloat16 and torch.isfinite(t).all()


def test_layer_native_vs_library_gemm_error_table():
    """The rule for the headline mode: against the fp64 oracle, 'native' may be at most 2x
    the plain-fp32 path's error on y, dx and all four parameter gradients."""
    import numpy as np
    from ampnet_amd import AMPConv
    from oracle.ampconv_numpy import AMPConvOracle
    dev = _dev()
    torch.manual_seed(5)
    N, E, L, D, H = 400, 4800, 20, 256, 8
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(6)
    x, dy = torch.randn(N, L * D, generator=g), torch.randn(N, L * D, generator=g)
    ei = torch.randint(0, N, (2, E), generator=g)
    m = layer.multi_head_attention
    o = AMPConvOracle(*(p.detach().cpu().double().numpy() for p in (m.in_proj_weight, m.in_proj_bias,
                                                                    m.out_proj.weight, m.out_proj.bias)), H)
    y_ref, _ = o.forward(x.double().numpy(), ei.numpy())
    refs = (y_ref,) + tuple(o.backward(dy.double().numpy()))
    errs = {}
    for mode in ('native', 'fp32'):
        layer.gemm_precision = mode
        layer.zero_grad(set_to_none=True)
        xg = x.to(dev).requires_grad_(True)
        y = layer(xg, ei.to(dev))
        y.backward(dy.to(dev))
        got = (y, xg.grad, m.in_proj_weight.grad, m.in_proj_bias.grad, m.out_proj.weight.grad, m.out_proj.bias.grad)
        errs[mode] = [float(np.abs(t.detach().cpu().double().numpy() - r).max() / np.abs(r).max())
                      for t, r in zip(got, refs)]
    for en, ef in zip(errs['native'], errs['fp32']):
        assert en <= 2 * ef + 2e-7, errs
