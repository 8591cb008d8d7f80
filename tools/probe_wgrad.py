"""Developer probe: proj_wgrad timing by operand shape / stride (why is the 768-column case slow?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ampnet_amd.conv import functional as F_
from tools.bench_proj import timeit
M, D = 20_000_000, 256
dev = torch.device('cuda:0')
x = torch.randn(M, D, device=dev)
qkv = torch.empty(M, 3 * D, device=dev)
for r0 in range(0, M, 1 << 20):
    qkv[r0:r0 + (1 << 20)] = torch.randn(min(1 << 20, M - r0), 3 * D, device=dev)
def run(name, a, b):
    dw = torch.empty(a.size(1), b.size(1), device=dev); cs = torch.empty(a.size(1), device=dev)
    t = timeit(lambda: F_.proj_wgrad(a, b, dw, cs), 3, 1)
    fl = 2 * M * a.size(1) * b.size(1)
    print(f'{name:50s} {t:8.2f} ms {fl / t / 1e9:7.1f} TF-equiv', flush=True)
run('A = qkv [M,768] lda 768, B = x', qkv, x)
run('A = qkv[:, :256] lda 768, B = x', qkv[:, :256], x)
run('A = qkv[:, :512] lda 768, B = x', qkv[:, :512], x)
run('A = x, B = x', x, x)
run('A = x, B = qkv[:, :256] ldb 768', x, qkv[:, :256])
run('A = x, B = qkv (Nb 768)', x, qkv)
