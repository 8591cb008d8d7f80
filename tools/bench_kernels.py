"""Developer micro-benchmark: time each C-ABI edge kernel on a synthetic uniform graph.

    python tools/bench_kernels.py [N E L D H] [--generic] [--bf16] [--hub] [--rmat] [--compact] [--absmax] [--planes] [--scaled]

--planes: the plane-format passes of ABI 106 (csrc/edge_mfma_f16x2.hip) on the same random tensors, converted to two
fp16 planes here (bounds 12 x the maxima, about what the a-priori bound of the projection gives).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ampnet_amd import _lib, EdgeCSR  # noqa: E402
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
    N, E, L, D, H = (int(x) for x in args) if len(args) == 5 else (100000, 1000000, 20, 256, 8)
    if '--generic' in sys.argv:
        os.environ['AMPCONV_FORCE_GENERIC'] = '1'
    dt = _lib.AMPCONV_F32
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dh = D // H
    torch.manual_seed(0)
    tdt = torch.bfloat16 if '--bf16' in sys.argv else torch.float32
    if tdt == torch.bfloat16:
        dt = _lib.AMPCONV_BF16
    pad = next((int(a.split('=')[1]) for a in sys.argv if a.startswith('--rowpad=')), 0)   # extra elements per token row
    qkv = torch.randn(N * L, 3 * D + pad, device=dev).to(tdt)
    dobar = torch.randn(N * L, D + pad, device=dev).to(tdt)
    ei = torch.randint(0, N, (2, E), device=dev)
    if '--rmat' in sys.argv:                      # BASELINE config 5's generator (N must be a power of two)
        import bench
        ei = bench.rmat_edges(N.bit_length() - 1, E, torch.Generator(device=dev).manual_seed(1234), dev)
    if '--sortdeg' in sys.argv:                   # relabel nodes by descending in+out degree (0 = the biggest hub)
        deg = torch.bincount(ei[0], minlength=N) + torch.bincount(ei[1], minlength=N)
        order = torch.argsort(deg, descending=True, stable=True)
        relabel = torch.empty(N, dtype=torch.int64, device=dev)
        relabel[order] = torch.arange(N, device=dev)
        ei = relabel[ei]
    if '--shuffle' in sys.argv:                   # relabel nodes by a random permutation
        relabel = torch.randperm(N, device=dev)
        ei = relabel[ei]
    if '--compact' in sys.argv:                   # relabel so that only nodes with an edge remain (N shrinks)
        ids = torch.unique(ei)
        relabel = torch.empty(N, dtype=torch.int64, device=dev)
        relabel[ids] = torch.arange(ids.numel(), device=dev)
        ei = relabel[ei]
        N = int(ids.numel())
        qkv, dobar = qkv[: N * L], dobar[: N * L]
        print('compacted to', N, 'nodes')
    if '--hub' in sys.argv:                       # 5 % of the edges end at node 3, 5 % start at node 5
        ei[1, : E // 20] = 3
        ei[0, E // 20: E // 10] = 5
    t0 = time.time()
    csr = EdgeCSR(ei, N)
    torch.cuda.synchronize()
    print(f'mode={dt} N={N} E={E} L={L} D={D} H={H}  csr build (cold) {1e3 * (time.time() - t0):.1f} ms')
    print(f'csr build {timeit(lambda: EdgeCSR(ei, N, validate=False)):.3f} ms')
    Qv, Kv, Vv = (F_._view(qkv, i * D, L, dh) for i in range(3))
    dOv = F_._view(dobar, 0, L, dh)
    # --layout=nhld / hnld: gathered inputs with contiguous per-(node, head) tiles (strided views, no kernel change)
    layout = next((a.split('=')[1] for a in sys.argv if a.startswith('--layout=')), 'nld')
    if layout != 'nld':
        src4 = [qkv[:, i * D:(i + 1) * D].reshape(N, L, H, dh) for i in range(3)] + [dobar.reshape(N, L, H, dh)]
        es = qkv.element_size()
        if layout == 'nhld':
            keep_in = [t.permute(0, 2, 1, 3).contiguous() for t in src4]          # [N, H, L, dh]
            views = [_lib.View(t.data_ptr(), H * L * dh, dh, L * dh) for t in keep_in]
        else:
            keep_in = [t.permute(2, 0, 1, 3).contiguous() for t in src4]          # [H, N, L, dh]
            views = [_lib.View(t.data_ptr(), L * dh, dh, N * L * dh) for t in keep_in]
        Qv, Kv, Vv, dOv = views
    print('input layout', layout)
    obar = torch.empty(N * L, D + pad, device=dev, dtype=tdt)
    dqkv = torch.empty(N * L, 3 * D + pad, device=dev, dtype=tdt)
    dQv, dKv, dVv = (F_._view(dqkv, i * D, L, dh) for i in range(3))
    st = torch.cuda.current_stream().cuda_stream
    R = L * D * qkv.element_size()

    keep = {}

    def hub(side, tiles):
        plan, n, ws = csr.hub_args(side, L, D, tiles)
        keep[side] = ws
        return plan, n, (ws.data_ptr() if ws is not None else None)

    def fwd():
        _lib.check(lib.ampconv_fwd_edge(Qv, Kv, Vv, csr.rowptr.data_ptr(), csr.col.data_ptr(), None,
                                        N, L, D, H, F_._view(obar, 0, L, dh), *hub('dst', 1), dt, st), 'fwd')

    # softmax statistics handed from the destination pass to the source pass (--no-stats: off)
    nstat = 0 if '--no-stats' in sys.argv else lib.ampconv_softmax_stats_bytes(E, L, D, H, dt)
    stats = torch.empty(nstat // 4, device=dev) if nstat else None
    spos = csr.csc_positions() if nstat else None
    sp = (spos.data_ptr(), stats.data_ptr()) if nstat else (None, None)
    print('softmax stats:', f'{nstat / 1e9:.2f} GB' if nstat else 'off')
    # --absmax: the backward passes record the maximum of what they write (fp32; include/ampconv.h out_absmax)
    amax_t = torch.zeros(1, device=dev) if '--absmax' in sys.argv else None
    amax = amax_t.data_ptr() if amax_t is not None else None

    def bwd_dst():
        _lib.check(lib.ampconv_bwd_edge_dst(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(),
                                            N, L, D, H, dQv, *hub('dst', 1), *sp, amax, dt, st), 'bwd_dst')

    def bwd_src():
        _lib.check(lib.ampconv_bwd_edge_src(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(),
                                            csr.cinv.data_ptr(), N, L, D, H, dKv, dVv, *hub('src', 2), sp[1],
                                            amax, dt, st), 'bwd_src')

    if '--planes' in sys.argv:
        import math

        def to_planes(t2, bound):                  # [M, W] fp32 -> the 128-byte slots of include/ampconv.h
            sc = 2.0 ** (14 - math.floor(math.log2(bound)))
            xs = (t2 * sc).view(t2.size(0), -1, 32)
            hi = xs.half()
            lo = (xs - hi.float()).half()
            return torch.cat([hi, lo], dim=2).contiguous().view(torch.float32).view(t2.size(0), -1)
        mq, mg = float(qkv.abs().max()), float(dobar.abs().max())
        bounds = torch.tensor([12 * mq, 12 * mg, mq, mg], device=dev)
        pq, pg = to_planes(qkv, 12 * mq), to_planes(dobar, 12 * mg)
        pQ, pK, pV = (F_._view(pq, i * D, L, dh) for i in range(3))
        pG = F_._view(pg, 0, L, dh)

        def fwd():
            _lib.check(lib.ampconv_fwd_edge_planes(pQ, pK, pV, csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L, D, H,
                                                   F_._view(obar, 0, L, dh), *hub('dst', 1), bounds.data_ptr(), st), 'fwd')

        def bwd_dst():
            _lib.check(lib.ampconv_bwd_edge_dst_planes(pQ, pK, pV, pG, csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L, D,
                                                       H, dQv, *hub('dst', 1), bounds.data_ptr(), *sp, amax, st), 'bwd_dst')

        def bwd_src():
            _lib.check(lib.ampconv_bwd_edge_src_planes(pQ, pK, pV, pG, csr.cscptr.data_ptr(), csr.crow.data_ptr(), N, L, D,
                                                       H, dKv, dVv, *hub('src', 2), bounds.data_ptr(), sp[1], amax, st), 'bwd_src')
        print('plane-format passes (fp16 planes, 16-bit matrix pipe)')

    if '--scaled' in sys.argv:                      # ABI 107: fp32 views + operand bounds (csrc/edge_block_x3.hip, two fp16 planes)
        mq, mg = float(qkv.abs().max()), float(dobar.abs().max())
        bounds = torch.tensor([mq, mg, mq, mg], device=dev)

        def fwd():
            _lib.check(lib.ampconv_fwd_edge_scaled(Qv, Kv, Vv, csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L, D, H,
                                                   F_._view(obar, 0, L, dh), *hub('dst', 1), bounds.data_ptr(), st), 'fwd')

        def bwd_dst():
            _lib.check(lib.ampconv_bwd_edge_dst_scaled(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L, D,
                                                       H, dQv, *hub('dst', 1), bounds.data_ptr(), *sp, amax, st), 'bwd_dst')

        def bwd_src():
            _lib.check(lib.ampconv_bwd_edge_src_scaled(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(),
                                                       csr.cinv.data_ptr(), N, L, D, H, dKv, dVv, *hub('src', 2),
                                                       bounds.data_ptr(), sp[1], amax, st), 'bwd_src')
        print('bound-carrying fp32-view passes (two fp16 planes split in the kernel, 16-bit matrix pipe)')

    for name, fn, nbytes, flops in (
            ('fwd_edge', fwd, (2 * E + 2 * N) * R, 4 * L * L * D * E),
            ('bwd_edge_dst', bwd_dst, (2 * E + 3 * N) * R, 6 * L * L * D * E),
            ('bwd_edge_src', bwd_src, (2 * E + 4 * N) * R, 8 * L * L * D * E)):
        ms = timeit(fn)
        print(f'{name:14s} {ms:9.3f} ms  {E / ms / 1e3:8.2f} M edges/s  '
              f'{nbytes / ms / 1e9:7.2f} TB/s alg  {flops / ms / 1e9:7.1f} TFLOP/s')


if __name__ == '__main__':
    main()
