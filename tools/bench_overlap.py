"""Developer experiment: do the fp32 projection GEMMs (matrix-pipe bound) and the edge kernels (HBM /
latency bound, matrix pipe ~50 % busy) overlap when issued on two HIP streams?

    python tools/bench_overlap.py [N E] [--native [--scaled]] [--planes] [--edges-first] [--cu-split=64,96,128]
Times (a) the two backward edge passes alone, (b) a set of GEMMs alone, (c) both concurrently.
--cu-split=K,...: the same with CU-MASKED streams (hipExtStreamCreateWithCUMask): the projections on K of the 256 CUs,
the edge passes on the other 256 - K (a power-limited GEMM on fewer CUs holds a higher clock, a
latency-bound gather on fewer CUs sees more HBM per CU)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ampnet_amd import _lib, EdgeCSR  # noqa: E402
from ampnet_amd.conv import functional as F_  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    N, E = (int(args[0]), int(args[1])) if len(args) == 2 else (100000, 1000000)
    L, D, H = 20, 256, 8
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dh = D // H
    torch.manual_seed(0)
    qkv = torch.randn(N * L, 3 * D, device=dev)
    dobar = torch.randn(N * L, D, device=dev)
    x = torch.randn(N * L, D, device=dev)
    w = torch.randn(D, D, device=dev)
    ei = torch.randint(0, N, (2, E), device=dev)
    csr = EdgeCSR(ei, N)
    Qv, Kv, Vv = (F_._view(qkv, i * D, L, dh) for i in range(3))
    dqkv = torch.empty(N * L, 3 * D, device=dev)
    dQv, dKv, dVv = (F_._view(dqkv, i * D, L, dh) for i in range(3))
    dOv = F_._view(dobar, 0, L, dh)
    nstat = lib.ampconv_softmax_stats_bytes(E, L, D, H, 0)
    stats = torch.empty(nstat // 4, device=dev)
    spos = csr.csc_positions()
    side = torch.cuda.Stream()
    out1 = torch.empty(N * L, D, device=dev)

    def edges():
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.ampconv_bwd_edge_dst(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L, D, H, dQv,
                                            None, 0, None, spos.data_ptr(), stats.data_ptr(), None, 0, st), 'dst')
        _lib.check(lib.ampconv_bwd_edge_src(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(),
                                            csr.cinv.data_ptr(), N, L, D, H, dKv, dVv, None, 0, None, stats.data_ptr(), None,
                                            0, st), 'src')

    if '--planes' in sys.argv:                       # round 5: the plane-format backward passes (16-bit matrix pipe, memory-bound)
        import math

        def to_planes(t2, bound):
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

        def edges():                                 # noqa: F811
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.ampconv_bwd_edge_dst_planes(pQ, pK, pV, pG, csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L, D, H,
                                                       dQv, None, 0, None, bounds.data_ptr(), spos.data_ptr(),
                                                       stats.data_ptr(), None, st), 'dst')
            _lib.check(lib.ampconv_bwd_edge_src_planes(pQ, pK, pV, pG, csr.cscptr.data_ptr(), csr.crow.data_ptr(), N, L, D, H,
                                                       dKv, dVv, None, 0, None, bounds.data_ptr(), stats.data_ptr(), None, st),
                       'src')
    native = '--native' in sys.argv                  # libampconv's own projection kernels (bf16 matrix cores)
    img = F_.proj_image(w) if native else None
    dwo, cso = torch.empty(D, D, device=dev), torch.empty(D, device=dev)

    scaled = '--scaled' in sys.argv                  # the projections' three-product mode (what large layers run)
    ax, ag = (F_.absmax(x), F_.absmax(dobar)) if scaled else (None, None)

    def gemms(k):
        for i in range(k):
            if not native:
                torch.mm(x, w, out=out1)             # [N L, D] x [D, D]: 2 N L D^2 flop each
            elif i % 2 == 0:
                F_.proj_rows(x, img, amax=ax)
            else:
                F_.proj_wgrad(x, dobar, dwo, cso, amax=(ax, ag) if scaled else None)    # the weight-gradient kernel: the same flop count

    def timed(fn, iters=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    flop = 2 * N * L * D * D
    splits = [int(v) for a in sys.argv if a.startswith('--cu-split=') for v in a.split('=')[1].split(',')]
    if splits:
        hip = ctypes.CDLL('libamdhip64.so')

        def masked_stream(cus):
            """Stream restricted to the CUs whose bit is set (bit i: the runtime deals the bits to the shader engines
            round-robin, so a prefix of the bits is spread evenly over the XCDs)."""
            words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if 32 * w + b in cus) for w in range(8)])
            st = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
            assert rc == 0, rc
            return torch.cuda.ExternalStream(st.value)

        t_e, t_g = timed(edges), timed(lambda: gemms(4))
        print(f'N={N} E={E}: whole chip: edge passes {t_e:.2f} ms | 4 projection launches {t_g:.2f} ms | serial {t_e + t_g:.2f} ms')
        for k in splits:
            sg, se = masked_stream(set(range(k))), masked_stream(set(range(k, 256)))

            def on(stream, fn):
                cur = torch.cuda.current_stream()
                stream.wait_stream(cur)
                with torch.cuda.stream(stream):
                    fn()
                cur.wait_stream(stream)
            t_em = timed(lambda: on(se, edges))
            t_gm = timed(lambda: on(sg, lambda: gemms(4)))

            def both():
                cur = torch.cuda.current_stream()
                sg.wait_stream(cur)
                se.wait_stream(cur)
                with torch.cuda.stream(sg):
                    gemms(4)
                with torch.cuda.stream(se):
                    edges()
                cur.wait_stream(sg)
                cur.wait_stream(se)
            t_b = timed(both)
            print(f'  projections on {k} CUs {t_gm:.2f} ms alone, edge passes on {256 - k} CUs {t_em:.2f} ms alone | '
                  f'concurrent {t_b:.2f} ms = {t_b / (t_e + t_g):.3f} x the whole-chip serial sum')
        return
    t_e = timed(edges)
    for k in (2, 4, 6):
        t_g = timed(lambda: gemms(k))

        def both():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            if '--edges-first' in sys.argv:
                edges()
                with torch.cuda.stream(side):
                    gemms(k)
            else:
                with torch.cuda.stream(side):
                    gemms(k)
                edges()
            cur.wait_stream(side)
        t_b = timed(both)
        print(f'N={N} E={E}: edge passes {t_e:.2f} ms | {k} GEMMs {t_g:.2f} ms ({k * flop / t_g / 1e9:.0f} TFLOP/s) | '
              f'concurrent {t_b:.2f} ms (serial sum {t_e + t_g:.2f}, saved {t_e + t_g - t_b:.2f} ms = '
              f'{100 * (t_e + t_g - t_b) / min(t_e, t_g):.0f} % of the shorter)')


if __name__ == '__main__':
    main()
