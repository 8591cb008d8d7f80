"""Developer probe: time per row of the bf16 in-projection (native and library) over the row count -- does the rate
depend on the footprint?   python tools/sweep_proj_rows.py [D]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ampnet_amd.conv import functional as F_  # noqa: E402
from bench_proj import timeit  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda:0')
torch.manual_seed(0)
w_in = (torch.randn(3 * D, D, device=dev) * 0.06).bfloat16()
b_in = (torch.randn(3 * D, device=dev) * 0.1).bfloat16()
img = F_.proj_image(w_in)
img_t = F_.proj_image(w_in, transpose=True)
for M in (1_000_000, 2_000_000, 4_000_000, 8_000_000, 12_000_000, 16_000_000, 20_000_000, 30_000_000, 41_943_040):
    x = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    for r0 in range(0, M, 1 << 20):
        n = min(1 << 20, M - r0)
        x[r0:r0 + n] = torch.randn(n, D, device=dev)
    t = timeit(lambda: F_.proj_rows(x, img, b_in), 3, 1)
    tl = timeit(lambda: torch.addmm(b_in, x, w_in.t()), 3, 1)
    q = F_.proj_rows(x, img, b_in)
    td = timeit(lambda: F_.proj_rows(q, img_t), 3, 1)
    print(f'M {M:9d}: qkv native {t:7.2f} ms = {M * 4 * D * 2 / t / 1e9:5.2f} TB/s ({t / M * 1e6:.3f} ns/row)   library {tl:7.2f} ms '
          f'= {M * 4 * D * 2 / tl / 1e9:5.2f} TB/s   dx native {td:7.2f} ms = {M * 4 * D * 2 / td / 1e9:5.2f} TB/s', flush=True)
    del x, q
    torch.cuda.empty_cache()
