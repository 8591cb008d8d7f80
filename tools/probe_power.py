"""Developer probe: is proj_rows power-limited?  Same launch on random / small-integer / zero operands
(MI355X_MICROARCH.md, DVFS give-back: zero operands raise the clock the chip holds under MFMA load)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ampnet_amd.conv import functional as F_
from tools.bench_proj import timeit
M, D = 8_000_000, 256
dev = torch.device('cuda:0')
w = torch.randn(3 * D, D, device=dev) * 0.06
for name, x, wt in (('random normal', torch.randn(M, D, device=dev), w),
                    ('bf16-exact values (planes 2, 3 zero)', torch.randn(M, D, device=dev).bfloat16().float(), w.bfloat16().float()),
                    ('zeros', torch.zeros(M, D, device=dev), torch.zeros_like(w))):
    img = F_.proj_image(wt)
    t = timeit(lambda: F_.proj_rows(x, img), 5, 2)
    print(f'{name:40s} {t:7.2f} ms  {2 * M * D * 3 * D * 6 / t / 1e9:6.0f} TF bf16 issued', flush=True)
