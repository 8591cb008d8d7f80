"""Fused AMPConv forward/backward on MI355X: projections as dense GEMMs on the
per-NODE rows, edge phase in hand-written HIP through the C ABI.

What it replaces in the reference (paths relative to /root/reference):
  src/ampnet/conv/amp_conv.py:24-26,28-51   propagate -> message -> mean
  torch functional.py:5785-5862             packed in-projection (per edge there, per node here)
  torch functional.py:6578-6601             scale, QK^T, softmax, PV, out-projection
Backward formulas: SURVEY.md A.2 (the reference relies on autograd).

Data layout in HBM (fp32):
  QKV   [N*L, 3D]   one packed projection; Q/K/V are column thirds (strided views)
  Obar  [N*L, D]    mean over in-edges of the per-edge attention output, pre out-proj
  dQKV  [N*L, 3D]   written by the two backward edge passes, consumed by one GEMM each
                    for dX and d(in_proj_weight)
"""
import contextlib
import ctypes
import os
import threading

import torch

from .. import _lib
from ..graph import EdgeCSR, _stream


def _view(buf2d, col_off, L, dh):
    """ampconv_view_t over columns [col_off, col_off + D) of a contiguous [rows*L, W] buffer."""
    assert buf2d.is_contiguous() and buf2d.dtype in (torch.float32, torch.bfloat16)
    W = buf2d.size(1)
    return _lib.View(buf2d.data_ptr() + buf2d.element_size() * col_off, L * W, W, dh)


def _ptr(t):
    return t.data_ptr() if t is not None else None


GEMM_PRECISIONS = ('native', 'fp32', 'bf16x3')
# hand the softmax statistics of the destination pass to the source pass (AMPCONV_SOFTMAX_STATS=0:
# both passes reduce their own; used by the tests to cross-check the two ways)
SOFTMAX_STATS = os.environ.get('AMPCONV_SOFTMAX_STATS', '1') != '0'


_GEMM_SWITCH_LOCK = threading.RLock()


@contextlib.contextmanager
def gemm_precision(mode):
    """How the dense fp32 projections run.  'fp32': plain fp32 MFMA GEMMs (rocBLAS, ~135-150
    TFLOP/s on MI355X).  'bf16x3': hipBLASLt's fp32 GEMM that splits each operand into two bf16
    planes and sums three bf16 MFMA products in fp32 (what ROCm serves as "TF32" on gfx950, which
    has no xf32 matrix instruction): ~2x the rate, element error ~5e-6 of the largest output
    instead of ~6e-7 (tools/bench_gemm.py).  Inputs, outputs and accumulation stay fp32; the
    edge kernels are not affected.

    torch's switches (allow_tf32, preferred BLAS library, HIPBLASLT_ALLOW_TF32) are PROCESS-GLOBAL:
    they are set for the duration of the block and restored, under a lock so that two layers (the
    autograd thread runs backward) never interleave their set/restore.  fp32 GEMMs issued by OTHER
    threads while a 'bf16x3' block is open run in that mode too -- keep the default 'fp32' if that
    matters (tests/test_gpu_parity.py::test_gemm_precision_is_restored pins the restore)."""
    if mode not in GEMM_PRECISIONS:
        raise ValueError(f'gemm precision must be one of {GEMM_PRECISIONS}, got {mode!r}')
    if mode in ('fp32', 'native'):      # 'native': libampconv's own projection kernels (proj_* below); what they
        yield                           # do not serve (bf16 storage, D % 4 != 0) runs as plain library GEMMs
        return
    with _GEMM_SWITCH_LOCK:
        prev_env = os.environ.get('HIPBLASLT_ALLOW_TF32')
        prev_tf32 = torch.backends.cuda.matmul.allow_tf32
        prev_lib = torch.backends.cuda.preferred_blas_library()
        os.environ['HIPBLASLT_ALLOW_TF32'] = '1'
        torch.backends.cuda.matmul.allow_tf32 = True
        torch.backends.cuda.preferred_blas_library('hipblaslt')
        try:
            yield
        finally:
            torch.backends.cuda.preferred_blas_library(prev_lib)
            torch.backends.cuda.matmul.allow_tf32 = prev_tf32
            if prev_env is None:
                os.environ.pop('HIPBLASLT_ALLOW_TF32', None)
            else:
                os.environ['HIPBLASLT_ALLOW_TF32'] = prev_env


def _tn_matmul(a, b, chunks=128):
    """a^T @ b for tall operands a [M, p], b [M, q] (the weight-gradient GEMMs, reduction over
    the N*L node-token rows).  rocBLAS runs the single [p, M] x [M, q] product at 65-95 TFLOP/s
    at M = 2e7; splitting the reduction into `chunks` batched GEMMs + one sum runs at
    ~150 TFLOP/s (measured on MI355X, tools/bench_dw.py) and sums pairwise, i.e. more accurately."""
    M = a.size(0)
    if M < (1 << 18):
        return a.t().mm(b)
    per = M // chunks
    main = per * chunks
    out = torch.bmm(a[:main].view(chunks, per, -1).transpose(1, 2), b[:main].view(chunks, per, -1)).sum(0)
    if main < M:
        out += a[main:].t().mm(b[main:])
    return out


# ---- the per-node projections in libampconv.so.  fp32 storage (csrc/proj_gemm.hip): operands scaled by a power of two
# and split into two fp16 planes, three partial products on v_mfma_f32_32x32x16_f16 (large operands; PROJ_SCALED below), or
# split exactly into three bf16 terms, six partial products on v_mfma_f32_32x32x16_bf16; fp32 accumulate.  bf16 storage
# (csrc/proj_gemm_bf16.hip): one product, fp32 accumulate, one rounding on the way out
NODE_LISTS = os.environ.get('AMPCONV_NODE_LISTS', '1') != '0'      # developer switch (A/B measurements)
# fp32 storage: two scaled fp16 planes, three products (include/ampconv.h "SCALED MODE"); '0': three bf16 planes, six
PROJ_SCALED = os.environ.get('AMPCONV_PROJ_SCALED', '1') != '0'
# ... for operands of at least this many elements: the mode costs five more (tiny) launches per layer call -- two
# maxima in the forward pass, two in the backward, one for the weights -- which a Cora-sized step (0.6 ms) feels and a
# GraphSAINT batch (24 ms) does not
PROJ_SCALED_MIN_ELEMENTS = 1 << 24


# fp32 storage, scaled mode, L <= 20, head width 32 or 16, self-attention layers (xq is xkv): the edge passes run on the
# 16-bit matrix pipe -- the in-projection and dObar = dY Wo leave their kernels as two fp16 planes of the scaled value
# (csrc/edge_mfma_f16x2.hip, include/ampconv.h "edge phase on fp16 PLANES"); '0': the fp32-MFMA edge kernels
EDGE_PLANES = os.environ.get('AMPCONV_EDGE_PLANES', '1') != '0'
# ONE power-of-two scale per tensor serves rows within this many binades of the tensor's maximum at fp32 grade; operands
# that have a head slot further down (ampconv_absmax_stats) take the exact kernels instead: six-product projections,
# fp32 edge passes.  The price of knowing is one 8-byte read-back per operand (x: cached per tensor version; dY: per step)
RANGE_LOG2 = 12
_STATS_CACHE = {}         # id(tensor) -> (weakref, _version, stats tensor, narrow, (data_ptr, shape))


def operand_stats(t2, key=None):
    """(stats, narrow): stats = device tensor [largest finite magnitude, smallest non-zero head-slot maximum] of the
    2-D fp32 tensor t2 (one pass, ampconv_absmax_stats), narrow = the whole tensor lies within 2^RANGE_LOG2 of its
    maximum (host bool: ONE 8-byte read-back).  key: the tensor OBJECT the caller was handed (t2 may be a view of it);
    the result is remembered for as long as that object lives and its version counter stands still -- the features of
    a full-graph run are measured once, not once per step."""
    import weakref
    if key is not None:
        hit = _STATS_CACHE.get(id(key))
        if (hit is not None and hit[0]() is key and hit[1] == key._version and hit[4] == (key.data_ptr(), tuple(key.shape))
                and hit[2].device == t2.device):
            return hit[2], hit[3]
    lib = _lib.load()
    t2 = _aligned(t2)
    st = torch.empty(2, dtype=torch.float32, device=t2.device)
    _lib.check(lib.ampconv_absmax_stats(t2.data_ptr(), t2.stride(0), t2.size(0), t2.size(1), st.data_ptr(), _stream()),
               'ampconv_absmax_stats')
    amax, smin = st.tolist()
    narrow = not (smin * float(1 << RANGE_LOG2) < amax)          # (an all-zero tensor: inf < 0 is false -> narrow)
    if key is not None:
        if len(_STATS_CACHE) >= 16:
            for k in [k for k, v in _STATS_CACHE.items() if v[0]() is None] or list(_STATS_CACHE)[:8]:
                _STATS_CACHE.pop(k, None)
        _STATS_CACHE[id(key)] = (weakref.ref(key), key._version, st, narrow, (key.data_ptr(), tuple(key.shape)))
    return st, narrow


def planes_to_f32(buf2d, bound, dh=32):
    """fp32 copy of a projection buffer held in the plane format with slots of `dh` channels (side outputs, fall-backs)."""
    out = torch.empty_like(buf2d)
    _lib.check(_lib.load().ampconv_planes_to_f32(buf2d.data_ptr(), buf2d.stride(0), buf2d.size(0), buf2d.size(1), dh,
                                                 bound.data_ptr(), out.data_ptr(), out.stride(0), _stream()),
               'ampconv_planes_to_f32')
    return out


def proj_out_bound(W, transpose, bias, amax, out):
    """out[0] = amax * (largest absolute row sum of B) + max |bias|, B = W (transpose: W^T): the scale source of a
    plane-format projection output, on the device."""
    R, C = W.shape
    N, K = (C, R) if transpose else (R, C)
    sn, sk = (1, W.stride(0)) if transpose else (W.stride(0), 1)
    _lib.check(_lib.load().ampconv_proj_out_bound(W.data_ptr(), sn, sk, N, K, _ptr(bias), amax.data_ptr(), out.data_ptr(),
                                                  _stream()), 'ampconv_proj_out_bound')


def proj_rows_planes(a2, image, bound, bias=None, rowptr=None, L=0, row_scale=0, amax=None, out_amax=None, amax_col0=0,
                     dh=32):
    """proj_rows whose output leaves as two fp16 planes of value * 2^e(bound) in the 4 dh-byte head slots of the fp32 buffer
    (include/ampconv.h; dh = 32 or 16).  row_scale = 1 (with rowptr): rows are divided by their node's in-degree (0 for none)."""
    lib = _lib.load()
    img, N, K, wdt = image
    assert a2.dim() == 2 and a2.size(1) == K and a2.stride(1) == 1 and a2.dtype == wdt == torch.float32
    a2 = _aligned(a2)
    out = torch.empty(a2.size(0), N, dtype=wdt, device=a2.device)
    _lib.check(lib.ampconv_proj_rows_planes(a2.data_ptr(), a2.stride(0), a2.size(0), K, img.data_ptr(), N, _ptr(bias),
                                            _ptr(rowptr), L, row_scale, out.data_ptr(), out.stride(0), amax.data_ptr(),
                                            bound.data_ptr(), _ptr(out_amax), amax_col0, dh, _stream()),
               'ampconv_proj_rows_planes')
    return out


# Sequences of at most 4 tokens stay on the short-sequence kernels (csrc/edge_small.hip: one wave per ROW, no tile padding):
# the plane-format kernels pad every sequence to 20-token tiles (config 3's L = 4 sweep: 4.0 ms per step there, 14.2 here)
PLANES_MIN_L = 5


def planes_ok(L, D, H, shared):
    """Does the plane-format edge phase serve this layer call?  (Only with the scaled projections; the caller checks.)"""
    return bool(EDGE_PLANES and shared and D % 128 == 0 and L >= PLANES_MIN_L
                and _lib.load().ampconv_planes_supported(L, D, H))


def scaled_views_ok(L, D, H, shared):
    """Do the bound-carrying fp32 entry points (include/ampconv.h, ampconv_*_edge_scaled: the workgroup-per-unit shapes --
    L <= 64, even head widths up to 64 outside the one-wave-per-unit kernels' L <= 20 x {16, 32} -- e.g. the AMPGCN class
    defaults L = 40, D = 100, H = 2) serve this layer call?  Only with the scaled projections and
    the statistics hand-off; the caller checks the former."""
    return bool(EDGE_PLANES and SOFTMAX_STATS and shared and L >= PLANES_MIN_L and _lib.load().ampconv_scaled_supported(L, D, H))


def absmax(t2, out=None, reset=False):
    """Largest finite magnitude of a 2-D tensor with contiguous rows, as a one-element device tensor (no host sync):
    the scale source of the fp32 projections' two-plane mode.  `out`: merge into an existing maximum (reset: zero it
    first)."""
    lib = _lib.load()
    t2 = _aligned(t2)
    if out is None:
        out, reset = torch.empty(1, dtype=torch.float32, device=t2.device), True
    _lib.check(lib.ampconv_absmax(t2.data_ptr(), t2.stride(0), t2.size(0), t2.size(1), _code(t2.dtype), out.data_ptr(),
                                  1 if reset else 0, _stream()), 'ampconv_absmax')
    return out


def _zero(scalar):
    """Zero a one-element fp32 device tensor on the current stream (the receiving end of proj_rows' out_amax)."""
    _lib.check(_lib.load().ampconv_absmax(None, 0, 0, 0, _lib.AMPCONV_F32, scalar.data_ptr(), 1, _stream()),
               'ampconv_absmax')


def _code(dtype):
    return _lib.AMPCONV_BF16 if dtype == torch.bfloat16 else _lib.AMPCONV_F32


def proj_native(gemm, dtype, D):
    """Does the 'native' mode serve this layer?  fp32 storage with embed_dim a multiple of 4, bf16 storage with a
    multiple of 8 (rows move in 16-byte pieces)."""
    return (gemm == 'native' and dtype in (torch.float32, torch.bfloat16)
            and bool(_lib.load().ampconv_proj_supported(D, D, _code(dtype))))


def _aligned(t):
    """Rows in 16-byte pieces: a view whose first element is not 16-byte aligned (a slice of a larger tensor) is copied."""
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def proj_images(jobs):
    """MFMA-fragment images of weights, ALL in one launch: `jobs` = [(W, transpose), ...] (at most 8), W [R, C]
    with contiguous rows, all of one dtype; an image of W itself serves out = in @ W^T (the forward direction of
    nn.Linear), of W^T serves out = in @ W (its input gradient).  Returns [(image bytes, N, K, dtype), ...]."""
    lib = _lib.load()
    descs, outs, off = (_lib.WeightImage * len(jobs))(), [], 0
    dims = []
    wdt = jobs[0][0].dtype
    code = _code(wdt)
    for W, transpose in jobs:
        assert W.dim() == 2 and W.stride(1) == 1 and W.dtype == wdt
        R, C = W.shape
        N, K = (C, R) if transpose else (R, C)
        dims.append((N, K, lib.ampconv_proj_weight_image_bytes(N, K, code)))
    buf = torch.empty(sum(d[2] for d in dims), dtype=torch.uint8, device=jobs[0][0].device)
    for i, ((W, transpose), (N, K, nb)) in enumerate(zip(jobs, dims)):
        sn, sk = (1, W.stride(0)) if transpose else (W.stride(0), 1)
        img = buf[off:off + nb]
        descs[i] = _lib.WeightImage(W.data_ptr(), sn, sk, N, K, img.data_ptr())
        outs.append((img, N, K, wdt))
        off += nb
    _lib.check(lib.ampconv_proj_weight_images(len(jobs), ctypes.cast(descs, ctypes.c_void_p), code, _stream()),
               'ampconv_proj_weight_images')
    return outs


def proj_image(W, transpose=False):
    return proj_images([(W, transpose)])[0]


def proj_rows(a2, image, bias=None, rowptr=None, L=0, nodes=None, out=None, amax=None, out_amax=None):
    """out[M, N] = (a2[M, K] @ B^T + bias) * [node of the row has an in-edge]  (mask only with rowptr).
    nodes = (ids, count[, ptr]) (EdgeCSR.active_nodes): only the L rows of each listed node are computed and written
    (bf16; the other rows of `out` keep what they hold: uninitialised unless the caller passes `out`).
    amax (fp32 storage): one-element tensor, the operand's largest finite magnitude or a bound of it (absmax) -> the
    two-plane scaled kernels; out_amax: a zeroed one-element tensor that receives the output's."""
    lib = _lib.load()
    img, N, K, wdt = image
    assert a2.dim() == 2 and a2.size(1) == K and a2.stride(1) == 1 and a2.dtype == wdt
    assert bias is None or bias.dtype == wdt
    a2 = _aligned(a2)
    if out is None:
        out = torch.empty(a2.size(0), N, dtype=wdt, device=a2.device)
    assert out.shape == (a2.size(0), N) and out.stride(1) == 1 and out.dtype == wdt
    ids, cnt = (nodes[0].data_ptr(), nodes[1]) if nodes is not None else (None, 0)
    _lib.check(lib.ampconv_proj_rows(a2.data_ptr(), a2.stride(0), a2.size(0), K, img.data_ptr(), N, _ptr(bias),
                                     _ptr(rowptr), L, out.data_ptr(), out.stride(0), ids, cnt, _ptr(amax), _ptr(out_amax),
                                     _code(wdt), _stream()), 'ampconv_proj_rows')
    return out


def proj_wgrad(a2, b2, dw, colsum=None, rowptr=None, L=0, nodes=None, amax=None):
    """dw[Na, Nb] = (mask * a2)^T @ b2 and colsum[Na] = column sums of mask * a2, into caller-owned (views of)
    contiguous tensors of the inputs' dtype; reduction over the rows in fixed slices (bitwise reproducible).
    bf16: the mask acts on the column sums only (include/ampconv.h).  nodes: the sums run over the L rows of each
    listed node only (bf16).  amax = (a2's, b2's) largest finite magnitudes (fp32 storage: scaled two-plane kernels)."""
    lib = _lib.load()
    M, Na = a2.shape
    Nb = b2.size(1)
    assert b2.size(0) == M and dw.shape == (Na, Nb) and dw.is_contiguous() and a2.stride(1) == 1 and b2.stride(1) == 1
    assert a2.dtype == b2.dtype == dw.dtype and (colsum is None or colsum.dtype == a2.dtype)
    a2, b2 = _aligned(a2), _aligned(b2)
    code = _code(a2.dtype)
    ids, cnt = (nodes[0].data_ptr(), nodes[1]) if nodes is not None else (None, 0)
    nws = lib.ampconv_proj_wgrad_workspace_bytes(cnt * L if nodes is not None else M, Na, Nb, code)
    ws = torch.empty(max(nws, 16) // 4, dtype=torch.float32, device=a2.device)
    _lib.check(lib.ampconv_proj_wgrad(a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0), M, Na, Nb,
                                      _ptr(rowptr), L, dw.data_ptr(), _ptr(colsum), ws.data_ptr(), nws, ids, cnt,
                                      _ptr(amax[0]) if amax else None, _ptr(amax[1]) if amax else None, code,
                                      _stream()), 'ampconv_proj_wgrad')


def _zero_unlisted(t2, nodes, n_nodes, L):
    """Zero the rows of the nodes a list leaves out (its CSR-shaped `ptr`: include/ampconv.h, ampconv_active_nodes)."""
    lib = _lib.load()
    assert t2.is_contiguous()
    io = _lib.AMPCONV_BF16 if t2.dtype == torch.bfloat16 else _lib.AMPCONV_F32
    _lib.check(lib.ampconv_mask_rows(t2.data_ptr(), nodes[2].data_ptr(), n_nodes, L * t2.size(1), io, _stream()),
               'ampconv_mask_rows')


def node_lists(csr, dtype, L, native, shared):
    """The graph's node lists when this call can use them: bf16 projections of a self-attention layer (xq is xkv) with
    16 <= L <= 128 on a graph where a good share of the nodes has no edge (EdgeCSR.active_nodes), else None."""
    if not (NODE_LISTS and native and shared and dtype == torch.bfloat16 and 16 <= L <= 128):
        return None
    return csr.active_nodes()


def edge_forward(Q, K, V, csr, n_rows, L, D, H, out2d, qidx=None, dtype=_lib.AMPCONV_F32):
    lib = _lib.load()
    plan, nch, ws = csr.hub_args('dst', L, D, 1) if qidx is None else (None, 0, None)
    rc = lib.ampconv_fwd_edge(Q, K, V, csr.rowptr.data_ptr(), csr.col.data_ptr(), _ptr(qidx),
                              n_rows, L, D, H, _view(out2d, 0, L, D // H), plan, nch, _ptr(ws),
                              dtype, _stream())
    _lib.check(rc, 'ampconv_fwd_edge')


class AMPConvFunction(torch.autograd.Function):
    """y = mask(deg>0) * (mean_in-edges(attention) Wo^T + bo), differentiable in
    xq, xkv and the four nn.MultiheadAttention parameters.

    `xq` supplies the query (destination) rows and `xkv` the key/value (source)
    rows; AMPConv.forward passes the same tensor for both, AMPConv.message passes
    the pre-gathered x_i / x_j with an identity graph."""

    @staticmethod
    def forward(ctx, xq, xkv, w_in, b_in, w_out, b_out, csr, num_heads, shared, dtype=_lib.AMPCONV_F32,
                gemm='fp32'):
        lib = _lib.load()
        D = w_out.size(0)
        H = int(num_heads)
        dh = D // H
        L = xq.size(1) // D
        if w_in.dtype != xq.dtype or w_out.dtype != xq.dtype:
            raise ValueError(f'inputs are {xq.dtype} but the parameters are {w_in.dtype}: storage is all float32 or all '
                             f'bfloat16 (layer.to(torch.bfloat16))')
        if xq.dtype == torch.bfloat16:                   # bf16 storage: one mode only
            dtype = _lib.AMPCONV_BF16
            if not ((dh in (16, 32) and L <= 20) or (dh % 2 == 0 and dh <= 64 and L <= 64)):
                raise ValueError(f'bf16 storage is implemented for head dimensions 32 and 16 with at most 20 tokens per '
                                 f'node (csrc/edge_mfma_bf16.hip; BASELINE configs 5 and 3) and for even head dimensions '
                                 f'up to 64 with at most 64 tokens (csrc/edge_block_x3.hip: the workgroup-per-unit kernels on bf16 rows); '
                                 f'got head dimension {dh}, {L} tokens: use float32 for this shape')
        Nq, Nk = xq.size(0), xkv.size(0)
        xq2 = xq.contiguous().view(Nq * L, D)
        native = proj_native(gemm, xq.dtype, D)
        imgs = None
        with torch.cuda.device(xq.device), gemm_precision(gemm):
            if native:      # every weight image this call and its backward need, in one launch
                ws = [w_in, w_out] if shared else [w_in[:D], w_in[D:], w_out]
                imgs = proj_images([(w, False) for w in ws] + [(w, True) for w in ws])
            lists = node_lists(csr, xq.dtype, L, native, shared and Nq == csr.num_nodes)
            # fp32 storage: operand maxima for the scaled two-plane products, device-side.  am = [x (query side), the
            # in-projection's output (K | V side: an upper bound of |Obar|, a mean of convex combinations of V rows),
            # x (key/value side)]; inputs are measured by one pass, outputs recorded by the product that writes them
            # The scaled mode serves operands that lie within 2^RANGE_LOG2 of their maximum (operand_stats: one pass, one
            # 8-byte read-back, remembered per tensor version); anything wider takes the exact six-product kernels.
            am = bounds = None
            planes = scaledv = False
            xkv2 = xq2 if shared else xkv.contiguous().view(Nk * L, D)
            # (the read-back rules the mode out while a HIP graph is being recorded: ampnet_amd/graphed.py is for small graphs)
            if (native and PROJ_SCALED and xq.dtype == torch.float32 and xq2.numel() >= PROJ_SCALED_MIN_ELEMENTS
                    and not torch.cuda.is_current_stream_capturing()):
                st, narrow = operand_stats(xq2, key=xq)
                stk = None
                if narrow and not shared:
                    stk, narrow = operand_stats(xkv2, key=xkv)
                if narrow:
                    planes = planes_ok(L, D, H, shared)
                    scaledv = not planes and scaled_views_ok(L, D, H, shared)
                    # device scalars of the plane format (include/ampconv.h): [bound of Q|K|V, bound of dObar, recorded
                    # max |V|, recorded max |dObar|]; max |V| doubles as the operand maximum of Obar (am[1]).
                    # fp32 views with bounds (`scaledv`): the in-projection records max |Q|K|V| into bounds[0], which serves
                    # as all of: the tensor's scale, the bound of |V|, and the operand maximum of Obar
                    bounds = torch.zeros(4, dtype=torch.float32, device=xq.device) if planes or scaledv else None
                    am = [st[0:1], bounds[2:3] if planes else bounds[0:1] if scaledv else
                          torch.zeros(1, dtype=torch.float32, device=xq.device), None if shared else stk[0:1]]
            sl = (lambda i: am[i]) if am is not None else (lambda i: None)
            if planes:
                # Q | K | V leave the in-projection as two fp16 planes scaled by a bound known before the product runs
                # (max |x| times the largest absolute row sum of W, plus max |b|); the recorded maximum covers the V third
                # only: it bounds Obar, a mean of convex combinations of V rows
                proj_out_bound(w_in, False, b_in, am[0], bounds[0:1])
                qkv = proj_rows_planes(xq2, imgs[0], bounds[0:1], b_in, amax=am[0], out_amax=am[1], amax_col0=2 * D, dh=dh)
                Qv, Kv, Vv = (_view(qkv, i * D, L, dh) for i in range(3))
                kv = None
            elif shared:
                # (node lists: Q rows matter for nodes with in-edges, K / V rows for nodes with out-edges; the rows of
                # nodes with neither are never read by an edge pass and stay unwritten)
                qkv = (proj_rows(xq2, imgs[0], b_in, L=L, nodes=lists and lists['any'], amax=sl(0), out_amax=sl(1))
                       if native else torch.addmm(b_in, xq2, w_in.t()))              # [N*L, 3D]
                Qv, Kv, Vv = (_view(qkv, i * D, L, dh) for i in range(3))
                kv = None
            else:
                if native:
                    qkv = proj_rows(xq2, imgs[0], b_in[:D], amax=sl(0))
                    kv = proj_rows(xkv2, imgs[1], b_in[D:], amax=sl(2), out_amax=sl(1))
                else:
                    qkv = torch.addmm(b_in[:D], xq2, w_in[:D].t())         # [Nq*L, D]
                    kv = torch.addmm(b_in[D:], xkv2, w_in[D:].t())         # [Nk*L, 2D]
                Qv = _view(qkv, 0, L, dh)
                Kv, Vv = _view(kv, 0, L, dh), _view(kv, D, L, dh)
            obar = torch.empty(Nq * L, D, dtype=xq.dtype, device=xq.device)
            if planes:
                plan, nch, ws = csr.hub_args('dst', L, D, 1)
                _lib.check(lib.ampconv_fwd_edge_planes(Qv, Kv, Vv, csr.rowptr.data_ptr(), csr.col.data_ptr(), Nq, L, D, H,
                                                       _view(obar, 0, L, dh), plan, nch, _ptr(ws), bounds.data_ptr(),
                                                       _stream()), 'ampconv_fwd_edge_planes')
            elif scaledv:
                bounds[2:3].copy_(bounds[0:1])
                plan, nch, ws = csr.hub_args('dst', L, D, 1)
                _lib.check(lib.ampconv_fwd_edge_scaled(Qv, Kv, Vv, csr.rowptr.data_ptr(), csr.col.data_ptr(), Nq, L, D, H,
                                                       _view(obar, 0, L, dh), plan, nch, _ptr(ws), bounds.data_ptr(),
                                                       _stream()), 'ampconv_fwd_edge_scaled')
            else:
                edge_forward(Qv, Kv, Vv, csr, Nq, L, D, H, obar, dtype=dtype)
            if lists:       # the rows of the nodes with in-edges; the others are zeroed without being read
                y = proj_rows(obar, imgs[1], b_out, L=L, nodes=lists['in'])
                _zero_unlisted(y, lists['in'], Nq, L)
            elif native:    # bias and the in-degree mask (rows nobody sends to stay exactly 0) in the epilogue
                y = proj_rows(obar, imgs[1 if shared else 2], b_out, csr.rowptr, L, amax=sl(1))
            else:
                y = torch.addmm(b_out, obar, w_out.t())
                io = _lib.AMPCONV_BF16 if y.dtype == torch.bfloat16 else _lib.AMPCONV_F32
                rc = lib.ampconv_mask_rows(y.data_ptr(), csr.rowptr.data_ptr(), Nq, L * D, io, _stream())
                _lib.check(rc, 'ampconv_mask_rows')
        ctx.set_materialize_grads(False)     # no zero-filled [N*L, 3D] gradient for the qkv side output
        ctx.save_for_backward(xq2, xkv2, w_in, w_out, qkv, kv, obar)
        ctx.csr, ctx.dims, ctx.shared, ctx.dtype, ctx.gemm = csr, (Nq, Nk, L, D, H), shared, dtype, gemm
        ctx.lists, ctx.amax, ctx.bounds, ctx.plane_format = lists, am, bounds, planes
        ctx.images_t = imgs[len(imgs) // 2:] if imgs else None     # the transposed images, for the input gradients
        ctx.mark_non_differentiable(qkv)
        if kv is not None:
            ctx.mark_non_differentiable(kv)
        if planes:                          # plane format: the third output is what reads `qkv` back (planes_to_f32)
            ctx.mark_non_differentiable(bounds)
            return y.view(Nq, L * D), qkv, bounds
        return y.view(Nq, L * D), qkv, kv

    @staticmethod
    def backward(ctx, dy, _dqkv=None, _dkv=None):
        lib = _lib.load()
        if dy is None:
            return (None,) * 11
        xq2, xkv2, w_in, w_out, qkv, kv, obar = ctx.saved_tensors
        csr, shared = ctx.csr, ctx.shared
        Nq, Nk, L, D, H = ctx.dims
        dh = D // H
        dev = dy.device
        need_xq, need_xkv = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        with torch.cuda.device(dev), gemm_precision(ctx.gemm):
            dy2 = dy.contiguous().view(Nq * L, D)
            native = proj_native(ctx.gemm, dy2.dtype, D)
            # out-projection: rows with no in-edge contribute nothing (their obar is 0, and
            # the bias gradient masks them explicitly)
            lists, am, bounds = ctx.lists, ctx.amax, ctx.bounds
            # maxima of the gradients that are operands: [dY, dQ (shared: dQKV), dK | dV].  The scaled products serve dY
            # only if it lies within 2^RANGE_LOG2 of its maximum too (one pass + one read-back per step: gradients are
            # new every step); otherwise the whole backward pass takes the exact kernels
            ag = None
            if am is not None:
                stg, narrow = operand_stats(dy2)
                if narrow:
                    ag = [stg[0:1], torch.zeros(1, dtype=torch.float32, device=dev), None]
                else:
                    am = None
            planes = ctx.plane_format and ag is not None
            scaledv = bounds is not None and not ctx.plane_format and ag is not None
            if ctx.plane_format and not planes:           # the saved projections as fp32 for the exact edge passes
                qkv = planes_to_f32(qkv, bounds[0:1], dh)
            sl = (lambda t, i: t[i]) if am is not None else (lambda t, i: None)
            pair = (lambda a, b: (a, b)) if am is not None else (lambda a, b: None)
            if native:
                dw_out = torch.empty_like(w_out)
                db_out = torch.empty(D, dtype=dy2.dtype, device=dev)
                if lists:   # the listed nodes ARE the ones that pass the mask; dObar is read for destinations only
                    proj_wgrad(dy2, obar, dw_out, db_out, L=L, nodes=lists['in'])
                    dobar = proj_rows(dy2, ctx.images_t[-1], L=L, nodes=lists['in'])
                elif planes:    # dObar / in-degree as two fp16 planes (the edge passes then carry no per-edge weight)
                    proj_wgrad(dy2, obar, dw_out, db_out, csr.rowptr, L, amax=pair(sl(ag, 0), sl(am, 1)))
                    proj_out_bound(w_out, True, None, ag[0], bounds[1:2])
                    bounds[3:4].zero_()
                    dobar = proj_rows_planes(dy2, ctx.images_t[-1], bounds[1:2], rowptr=csr.rowptr, L=L, row_scale=1,
                                             amax=ag[0], out_amax=bounds[3:4], dh=dh)
                elif scaledv:   # fp32 dObar, its maximum recorded by the product that writes it (bounds[1]; [3]: a copy)
                    proj_wgrad(dy2, obar, dw_out, db_out, csr.rowptr, L, amax=pair(sl(ag, 0), sl(am, 1)))
                    bounds[1:2].zero_()
                    dobar = proj_rows(dy2, ctx.images_t[-1], amax=ag[0], out_amax=bounds[1:2])
                    bounds[3:4].copy_(bounds[1:2])
                else:
                    proj_wgrad(dy2, obar, dw_out, db_out, csr.rowptr, L, amax=pair(sl(ag, 0), sl(am, 1)))
                    dobar = proj_rows(dy2, ctx.images_t[-1], amax=sl(ag, 0))
            else:
                scratch = torch.empty((1 + _lib.COLSUM_BLOCKS) * D, dtype=torch.float32, device=dev)
                io = _lib.AMPCONV_BF16 if dy2.dtype == torch.bfloat16 else _lib.AMPCONV_F32
                rc = lib.ampconv_masked_colsum(dy2.data_ptr(), csr.rowptr.data_ptr(), Nq, L, D,
                                               scratch.data_ptr(), io, _stream())
                _lib.check(rc, 'ampconv_masked_colsum')
                db_out = scratch[:D].to(dy2.dtype)
                dw_out = _tn_matmul(dy2, obar)
                dobar = dy2.mm(w_out)                                      # [Nq*L, D]
            dOv = _view(dobar, 0, L, dh)
            if shared:
                dqkv = torch.empty(Nq * L, 3 * D, dtype=dy2.dtype, device=dev)
                Qv, Kv, Vv = (_view(qkv, i * D, L, dh) for i in range(3))
                dQv, dKv, dVv = (_view(dqkv, i * D, L, dh) for i in range(3))
                dkv = None
            else:
                dqkv = torch.empty(Nq * L, D, dtype=dy2.dtype, device=dev)
                dkv = torch.empty(Nk * L, 2 * D, dtype=dy2.dtype, device=dev)
                Qv, Kv, Vv = _view(qkv, 0, L, dh), _view(kv, 0, L, dh), _view(kv, D, L, dh)
                dQv, dKv, dVv = _view(dqkv, 0, L, dh), _view(dkv, 0, L, dh), _view(dkv, D, L, dh)
            # softmax statistics (normaliser, delta) per edge: a by-product of the destination
            # pass that saves the source pass its cross-lane reductions (include/ampconv.h)
            stats = spos = None
            nstat = lib.ampconv_softmax_stats_bytes(csr.num_edges, L, D, H, ctx.dtype) if SOFTMAX_STATS else 0
            if nstat:
                stats = torch.empty(nstat // 4, dtype=torch.float32, device=dev)
                spos = csr.csc_positions()
            plan, nch, ws = csr.hub_args('dst', L, D, 1)
            if planes:
                # both passes on the 16-bit matrix pipe; each records the maximum of what it writes (ag[1]: all of dQKV)
                _lib.check(lib.ampconv_bwd_edge_dst_planes(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(),
                                                           Nq, L, D, H, dQv, plan, nch, _ptr(ws), bounds.data_ptr(),
                                                           _ptr(spos), _ptr(stats), ag[1].data_ptr(), _stream()),
                           'ampconv_bwd_edge_dst_planes')
                plan, nch, ws = csr.hub_args('src', L, D, 2)
                _lib.check(lib.ampconv_bwd_edge_src_planes(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(),
                                                           Nk, L, D, H, dKv, dVv, plan, nch, _ptr(ws), bounds.data_ptr(),
                                                           _ptr(stats), ag[1].data_ptr(), _stream()),
                           'ampconv_bwd_edge_src_planes')
            elif scaledv:
                _lib.check(lib.ampconv_bwd_edge_dst_scaled(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(),
                                                           Nq, L, D, H, dQv, plan, nch, _ptr(ws), bounds.data_ptr(),
                                                           _ptr(spos), _ptr(stats), ag[1].data_ptr(), _stream()),
                           'ampconv_bwd_edge_dst_scaled')
                plan, nch, ws = csr.hub_args('src', L, D, 2)
                _lib.check(lib.ampconv_bwd_edge_src_scaled(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(),
                                                           csr.cinv.data_ptr(), Nk, L, D, H, dKv, dVv, plan, nch, _ptr(ws),
                                                           bounds.data_ptr(), _ptr(stats), ag[1].data_ptr(), _stream()),
                           'ampconv_bwd_edge_src_scaled')
            else:
                # (scaled projections: the operand maximum of the two products that consume dQKV.  The destination pass
                # records the maximum of dQ as it stores; dK | dV: one pass below -- the fp32 source-pass kernels have no
                # register to spare for it (csrc/edge_mfma.hip); shared: one maximum, else dQ and dK | dV apart)
                rc = lib.ampconv_bwd_edge_dst(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(),
                                              Nq, L, D, H, dQv, plan, nch, _ptr(ws), _ptr(spos), _ptr(stats),
                                              _ptr(sl(ag, 1)), ctx.dtype, _stream())
                _lib.check(rc, 'ampconv_bwd_edge_dst')
                plan, nch, ws = csr.hub_args('src', L, D, 2)
                rc = lib.ampconv_bwd_edge_src(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(),
                                              csr.cinv.data_ptr(), Nk, L, D, H, dKv, dVv, plan, nch, _ptr(ws),
                                              _ptr(stats), None, ctx.dtype, _stream())
                _lib.check(rc, 'ampconv_bwd_edge_src')
                if am is not None:
                    if shared:
                        absmax(dqkv[:, D:], ag[1])                # merged into the maximum of dQ
                    else:
                        ag[2] = absmax(dkv)
            # in_proj_bias gradient without a pass over all of dQKV: softmax rows sum to 1, so the
            # column sum of dV over every source token equals the column sum of dObar over the rows
            # that receive messages, and dObar = dY Wo is linear in dY, so that sum is (masked column sum of
            # dY) Wo = db_out Wo: a [D] x [D, D] product instead of a second 20 GB reduction pass.  The K bias
            # shifts every score of a row equally, i.e. has gradient exactly 0 (torch's autograd returns ~1e-9 noise)
            db_v = None
            if shared and dy2.dtype == torch.float32 and not native:
                db_v = scratch[:D] @ w_out
            del dobar, stats
            if native:
                # weight and bias gradients in one pass each (the column sums ride on the rows the product reads
                # anyway: all three thirds of in_proj_bias.grad are the true sums, as autograd's are)
                dw_in = torch.empty_like(w_in)
                db_in = torch.empty(3 * D, dtype=dy2.dtype, device=dev)
                if lists:   # the dQKV rows of a node without any edge are zeros (both edge passes wrote them)
                    proj_wgrad(dqkv, xq2, dw_in, db_in, L=L, nodes=lists['any'])
                    dxq = dxkv = None
                    if need_xq:
                        dxq2 = proj_rows(dqkv, ctx.images_t[0], L=L, nodes=lists['any'])
                        _zero_unlisted(dxq2, lists['any'], Nq, L)
                        dxq = dxq2.view(Nq, L * D)
                elif shared:
                    proj_wgrad(dqkv, xq2, dw_in, db_in, amax=pair(sl(ag, 1), sl(am, 0)))
                    dxq = proj_rows(dqkv, ctx.images_t[0], amax=sl(ag, 1)).view(Nq, L * D) if need_xq else None
                    dxkv = None
                else:
                    proj_wgrad(dqkv, xq2, dw_in[:D], db_in[:D], amax=pair(sl(ag, 1), sl(am, 0)))
                    proj_wgrad(dkv, xkv2, dw_in[D:], db_in[D:], amax=pair(sl(ag, 2), sl(am, 2)))
                    dxq = proj_rows(dqkv, ctx.images_t[0], amax=sl(ag, 1)).view(Nq, L * D) if need_xq else None
                    dxkv = proj_rows(dkv, ctx.images_t[1], amax=sl(ag, 2)).view(Nk, L * D) if need_xkv else None
            elif shared:
                dw_in = _tn_matmul(dqkv, xq2)
                if db_v is not None:
                    db_in = torch.cat([dqkv[:, :D].sum(dim=0), torch.zeros_like(db_v), db_v])
                else:
                    db_in = dqkv.sum(dim=0)
                dxq = dqkv.mm(w_in).view(Nq, L * D) if need_xq else None
                dxkv = None
            else:
                dw_in = torch.cat([_tn_matmul(dqkv, xq2), _tn_matmul(dkv, xkv2)], dim=0)
                db_in = torch.cat([dqkv.sum(dim=0), dkv.sum(dim=0)])
                dxq = dqkv.mm(w_in[:D]).view(Nq, L * D) if need_xq else None
                dxkv = dkv.mm(w_in[D:]).view(Nk, L * D) if need_xkv else None
        return dxq, dxkv, dw_in, db_in, dw_out, db_out, None, None, None, None, None


def attention_weights(Qv, Kv, edge_index, L, D, H):
    """[E, L, L] head-averaged softmax weights in ORIGINAL edge order
    (amp_conv.py:39,43-47; torch functional.py:6604-6606)."""
    lib = _lib.load()
    E = edge_index.size(1)
    W = torch.empty(E, L, L, dtype=torch.float32, device=edge_index.device)
    with torch.cuda.device(edge_index.device):
        rc = lib.ampconv_attn_weights(Qv, Kv, edge_index.data_ptr(), E, L, D, H, W.data_ptr(),
                                      _lib.AMPCONV_F32, _stream())
    _lib.check(rc, 'ampconv_attn_weights')
    return W


class SegmentMeanFunction(torch.autograd.Function):
    """PyG aggr='mean' of an [E, F] message matrix over the dst-sorted CSR; differentiable in the
    messages (d msg[e] = d out[dst(e)] / in-degree(dst(e))), so the decomposed public path
    message() -> aggregate() trains like the fused propagate()."""

    @staticmethod
    def forward(ctx, msg, csr, index):
        lib = _lib.load()
        msg = msg.contiguous()
        N, F = csr.num_nodes, msg.size(1)
        out = torch.empty(N, F, dtype=torch.float32, device=msg.device)
        with torch.cuda.device(msg.device):
            rc = lib.ampconv_segment_mean(msg.data_ptr(), csr.rowptr.data_ptr(), csr.eperm.data_ptr(),
                                          N, F, out.data_ptr(), _stream())
        _lib.check(rc, 'ampconv_segment_mean')
        ctx.csr, ctx.index = csr, index
        return out

    @staticmethod
    def backward(ctx, dout):
        csr = ctx.csr
        deg = (csr.rowptr[1:] - csr.rowptr[:-1]).clamp(min=1).to(dout.dtype)
        return (dout / deg[:, None]).index_select(0, ctx.index), None, None


def segment_mean(msg, csr, index=None):
    """PyG aggr='mean' of an [E, F] message matrix over the dst-sorted CSR.  `index` (the int64
    destination of every message, original order) is needed only for the gradient."""
    if index is None:
        if msg.requires_grad and torch.is_grad_enabled():
            raise ValueError('segment_mean of messages that require grad needs `index`')
        index = torch.empty(0, dtype=torch.int64, device=msg.device)
    return SegmentMeanFunction.apply(msg, csr, index)
