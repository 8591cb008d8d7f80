"""Developer tool: torch TunableOp over the layer's dense GEMM shapes (rocBLAS + hipBLASLt solution sweep).

    python tools/tune_gemm.py [N_rows] [out.csv]

Times each shape with the library default first, then lets TunableOp pick the fastest solution and times again.
The result file can be replayed (tuning off) with PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0
PYTORCH_TUNABLEOP_FILENAME=<csv>.
"""
import os
import sys
import time

import torch

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
OUT = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/tunableop.csv'
D = 256
CH = 128


def t(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    x = torch.randn(M, D, device=dev)
    w3 = torch.randn(3 * D, D, device=dev) * 0.05
    b3 = torch.randn(3 * D, device=dev)
    w = torch.randn(D, D, device=dev) * 0.05
    b1 = torch.randn(D, device=dev)
    g3 = torch.randn(M, 3 * D, device=dev)
    per = M // CH
    shapes = {
        'qkv   addmm [M,256]x[256,768]': (lambda: torch.addmm(b3, x, w3.t()), 3),
        'out   addmm [M,256]x[256,256]': (lambda: torch.addmm(b1, x, w.t()), 1),
        'dx    mm    [M,768]x[768,256]': (lambda: g3.mm(w3), 3),
        'dobar mm    [M,256]x[256,256]': (lambda: x.mm(w), 1),
        'dw_in bmm   128x[768,per]x[per,256]': (lambda: torch.bmm(g3[:per * CH].view(CH, per, -1).transpose(1, 2),
                                                                  x[:per * CH].view(CH, per, -1)), 3),
        'dw_out bmm  128x[256,per]x[per,256]': (lambda: torch.bmm(x[:per * CH].view(CH, per, -1).transpose(1, 2),
                                                                  x[:per * CH].view(CH, per, -1)), 1),
    }
    unit = 2.0 * M * D * D
    base = {}
    for k, (fn, u) in shapes.items():
        base[k] = t(fn)
        print(f'default {k:40s} {base[k]:8.2f} ms  {u * unit / base[k] / 1e9:6.1f} TF', flush=True)
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_filename(OUT)
    torch.cuda.tunable.set_max_tuning_iterations(3)
    torch.cuda.tunable.set_max_tuning_duration(200)
    tot0 = tot1 = 0.0
    for k, (fn, u) in shapes.items():
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        print(f'tuned   {k:40s} (tuning took {time.time() - t0:.0f} s)', flush=True)
        ms = t(fn)
        print(f'tuned   {k:40s} {ms:8.2f} ms  {u * unit / ms / 1e9:6.1f} TF   ({base[k] / ms:.3f}x)', flush=True)
        tot0 += base[k]
        tot1 += ms
    print(f'sum default {tot0:.1f} ms   tuned {tot1:.1f} ms')
    torch.cuda.tunable.write_file()
    print(open(OUT).read() if os.path.exists(OUT) else 'no file ' + OUT)


if __name__ == '__main__':
    main()
