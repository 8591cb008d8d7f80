"""Dev: HBM rate of ampconv_gather_segment_sum (the edge phase of the softmax-free variant).
    python tools/bench_segment_sum.py [N E F]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ampnet_amd import EdgeCSR  # noqa: E402
from ampnet_amd.conv.linear import gather_segment_sum  # noqa: E402

N, E, F = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1_000_000, 10_000_000, 8192)
dev = torch.device('cuda:0')
rows = torch.randn(N, F, device=dev)
ei = torch.randint(0, N, (2, E), device=dev)
csr = EdgeCSR(ei, N)
for name, fn in (('mean over CSR', lambda: gather_segment_sum(rows, csr.rowptr, csr.col, N, mean=True)),
                 ('weighted sum over CSC', lambda: gather_segment_sum(rows, csr.cscptr, csr.crow, N, weights=csr.cinv))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f'N={N} E={E} F={F} {name:22s} {ms:8.3f} ms  {(E + N) * F * 4 / ms / 1e9:6.2f} TB/s  {E / ms / 1e3:7.1f} M edges/s')
