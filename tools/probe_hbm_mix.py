"""Developer probe: what does HBM give for the read / write mixes of the bf16 projections, with no arithmetic at all?
(torch elementwise kernels: fill = write only, copy = 1 R : 1 W, broadcast copy x -> [x, x, x] = 1 R : 3 W = the in-projection's
mix, sum of three column blocks = 3 R : 1 W = dX's mix)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402
from bench_proj import timeit  # noqa: E402

dev = torch.device('cuda:0')
M, D = 20_000_000, 256
x = torch.randn(M, D, device=dev).bfloat16()
y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
q = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
B = M * D * 2
for name, fn, nbytes in (
        ('fill (write only, 1 unit)', lambda: y.fill_(1.0), B),
        ('fill (write only, 3 units)', lambda: q.fill_(1.0), 3 * B),
        ('copy 1R:1W', lambda: y.copy_(x), 2 * B),
        ('broadcast copy 1R:3W (in-projection mix)', lambda: q.view(M, 3, D).copy_(x.view(M, 1, D).expand(M, 3, D)), 4 * B),
        ('column-block sum 3R:1W (dX mix)', lambda: torch.add(q[:, :D], q[:, D:2 * D], out=y).add_(q[:, 2 * D:]), 4 * B + 2 * B),
        ('read only (sum)', lambda: q.sum(), 3 * B)):
    t = timeit(fn, 5, 2)
    print(f'{name:44s} {t:7.2f} ms  {nbytes / t / 1e9:5.2f} TB/s', flush=True)
