"""Developer micro-benchmark of the node-phase projection kernels (csrc/proj_gemm.hip) against the
library GEMMs they replace, at the row count of a workload (default: cfg4, N*L = 2e7 rows, D = 256).

    python tools/bench_proj.py [rows D] [--iters=5] [--bf16] [--scaled] [--no-lib] [--only=rows|qkv|out|dx|dwin|dwo]
--scaled: fp32 storage through the two-plane scaled kernels (operand maxima by ampconv_absmax, timed separately)

Prints per GEMM: ms, fp32-equivalent TFLOP/s (2 M N K / t), bf16 MFMA TFLOP/s actually issued (x6),
and the max error against an fp64 product on a row sample."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ampnet_amd.conv import functional as F_  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    M, D = (int(args[0]), int(args[1])) if len(args) == 2 else (20_000_000, 256)
    iters = 5
    only = None
    lib = '--no-lib' not in sys.argv
    for a in sys.argv:
        if a.startswith('--iters='):
            iters = int(a.split('=')[1])
        if a.startswith('--only='):
            only = a.split('=')[1]
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    bf16 = '--bf16' in sys.argv                   # bf16 storage (csrc/proj_gemm_bf16.hip): one product, HBM-bound
    dt = torch.bfloat16 if bf16 else torch.float32
    scaled = '--scaled' in sys.argv and not bf16
    nprod = 1 if bf16 else (3 if scaled else 6)
    x = torch.empty(M, D, device=dev, dtype=dt)
    qkv = torch.empty(M, 3 * D, device=dev, dtype=dt)
    for r0 in range(0, M, 1 << 20):
        n = min(1 << 20, M - r0)
        x[r0:r0 + n] = torch.randn(n, D, device=dev)
        qkv[r0:r0 + n] = torch.randn(n, 3 * D, device=dev)
    w_in = (torch.randn(3 * D, D, device=dev) * 0.06).to(dt)
    b_in = (torch.randn(3 * D, device=dev) * 0.1).to(dt)
    w_out = (torch.randn(D, D, device=dev) * 0.06).to(dt)
    S = min(M, 4096)
    es = x.element_size()

    def report(name, t, flops, err=None, terr=None, nbytes=None):
        if t is None:
            return
        print(f'{name:34s} {t:8.2f} ms  {flops / t / 1e9:7.1f} TF  ({nprod * flops / t / 1e9:7.0f} TF bf16 issued)'
              + (f'  {nbytes / t / 1e9:6.2f} TB/s' if nbytes else '')
              + (f'  err {err:.2e}' + (f' (library {terr:.2e})' if terr else '') if err is not None else ''), flush=True)

    _timeit = timeit

    def timeit_lib(fn, n):
        return _timeit(fn, n) if lib else None

    def err(a, ref):
        return float((a.double() - ref).abs().max() / ref.abs().max())

    ax = aq = None
    if scaled:
        ax, aq = F_.absmax(x), F_.absmax(qkv)
        print(f'absmax x [M, D]: {timeit(lambda: F_.absmax(x), iters):.2f} ms   qkv [M, 3D]: {timeit(lambda: F_.absmax(qkv), iters):.2f} ms', flush=True)
        am_out = torch.zeros(1, device=dev)
    want = lambda name: only in (None, name) or (only == 'rows' and name in ('qkv', 'out', 'dx'))
    # forward in-projection [M, D] x [D, 3D]
    img = F_.proj_image(w_in)
    t = timeit(lambda: F_.proj_rows(x, img, b_in, amax=ax, out_amax=am_out if scaled else None), iters) if want('qkv') else None
    ref = x[:S].double() @ w_in.double().t() + b_in.double()
    report('qkv  native', t, 2 * M * D * 3 * D, err(F_.proj_rows(x[:S], img, b_in, amax=ax), ref), err(torch.addmm(b_in, x[:S], w_in.t()), ref),
           nbytes=M * 4 * D * es)
    report('qkv  library (addmm)', timeit_lib(lambda: torch.addmm(b_in, x, w_in.t()), iters), 2 * M * D * 3 * D, nbytes=M * 4 * D * es)
    if scaled and want('qkv') and D % 128 == 0:      # the same product leaving as two fp16 planes (the format of csrc/edge_mfma_f16x2.hip; embed_dim % 128 == 0)
        bound = torch.empty(1, device=dev)
        F_.proj_out_bound(w_in, False, b_in, ax, bound)
        t = timeit(lambda: F_.proj_rows_planes(x, img, bound, b_in, amax=ax, out_amax=am_out, amax_col0=2 * D), iters)
        back = F_.planes_to_f32(F_.proj_rows_planes(x[:S], img, bound, b_in, amax=ax), bound)
        report('qkv  native, plane output', t, 2 * M * D * 3 * D, err(back, ref), None and 0.0, nbytes=M * 4 * D * es)
        L = 20
        deg = torch.randint(1, 20, (M // L + 1,), device=dev)
        rowptr = torch.zeros(M // L + 2, dtype=torch.int32, device=dev)
        rowptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
        img_ot = F_.proj_image(w_out, transpose=True)
        F_.proj_out_bound(w_out, True, None, ax, bound)
        t = timeit(lambda: F_.proj_rows_planes(x[: (M // L) * L], img_ot, bound, rowptr=rowptr, L=L, row_scale=1, amax=ax, out_amax=am_out), iters)
        report('dObar native, plane output /deg', t, 2 * M * D * D, nbytes=M * 2 * D * es)
    # out-projection [M, D] x [D, D]
    img_o = F_.proj_image(w_out)
    report('out  native', timeit(lambda: F_.proj_rows(x, img_o, b_in[:D], amax=ax), iters) if want('out') else None, 2 * M * D * D, nbytes=M * 2 * D * es)
    report('out  library (addmm)', timeit_lib(lambda: torch.addmm(b_in[:D], x, w_out.t()), iters), 2 * M * D * D, nbytes=M * 2 * D * es)
    # dX = dQKV Win  [M, 3D] x [3D, D]
    img_t = F_.proj_image(w_in, transpose=True)
    t = timeit(lambda: F_.proj_rows(qkv, img_t, amax=aq), iters) if want('dx') else None
    ref = qkv[:S].double() @ w_in.double()
    report('dx   native', t, 2 * M * D * 3 * D, err(F_.proj_rows(qkv[:S], img_t, amax=aq), ref), err(qkv[:S].mm(w_in), ref), nbytes=M * 4 * D * es)
    report('dx   library (mm)', timeit_lib(lambda: qkv.mm(w_in), iters), 2 * M * D * 3 * D, nbytes=M * 4 * D * es)
    if only in ('rows', 'qkv', 'out', 'dx'):
        return
    # dW_in = dQKV^T X  [3D, M] x [M, D]
    dw, cs = torch.empty(3 * D, D, device=dev, dtype=dt), torch.empty(3 * D, device=dev, dtype=dt)
    t = timeit(lambda: F_.proj_wgrad(qkv, x, dw, cs, amax=(aq, ax) if scaled else None), iters) if want('dwin') else None
    Mr = min(M, 1 << 18)
    dwr, csr_ = torch.empty_like(dw), torch.empty_like(cs)
    F_.proj_wgrad(qkv[:Mr], x[:Mr], dwr, csr_, amax=(aq, ax) if scaled else None)
    ref = qkv[:Mr].double().t() @ x[:Mr].double()
    report('dWin native (+ colsum)', t, 2 * M * D * 3 * D, err(dwr, ref), err(qkv[:Mr].t().mm(x[:Mr]), ref), nbytes=M * 4 * D * es)
    report('dWin library (128-way bmm + sum)', timeit_lib(lambda: F_._tn_matmul(qkv, x), iters), 2 * M * D * 3 * D, nbytes=M * 4 * D * es)
    dwo, cso = torch.empty(D, D, device=dev, dtype=dt), torch.empty(D, device=dev, dtype=dt)
    x2 = qkv[:, :D].contiguous()
    report('dWo  native (+ colsum)', timeit(lambda: F_.proj_wgrad(x, x2, dwo, cso, amax=(ax, aq) if scaled else None), iters) if want('dwo') else None, 2 * M * D * D, nbytes=M * 2 * D * es)
    report('dWo  library', timeit_lib(lambda: F_._tn_matmul(x, x2), iters), 2 * M * D * D, nbytes=M * 2 * D * es)
    print('weight image (4 per step):', f'{timeit(lambda: F_.proj_image(w_in), 20):.4f} ms')


if __name__ == '__main__':
    main()
