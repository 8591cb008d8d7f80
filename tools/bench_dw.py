"""Dev: time weight-gradient GEMM formulations at cfg4 size (M = 20M rows)."""
import torch, time
dev = torch.device('cuda:0')
M, D = 20_000_000, 256
x = torch.randn(M, D, device=dev)
g = torch.randn(M, 3 * D, device=dev)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
ms, ref = t(lambda: g.t().mm(x)); print(f'g.t().mm(x)              {ms:8.2f} ms  {2*M*D*3*D/ms/1e9:6.1f} TF')
ms, r2 = t(lambda: x.t().mm(g).t()); print(f'(x.t().mm(g)).t()        {ms:8.2f} ms  err {(r2-ref).abs().max().item():.3e}')
for S in (64, 128, 256):
    def f(S=S):
        return torch.bmm(g.view(S, M // S, 3 * D).transpose(1, 2), x.view(S, M // S, D)).sum(0)
    ms, r = t(f); print(f'bmm S={S:4d} + sum         {ms:8.2f} ms  err {(r-ref).abs().max().item():.3e}')
g1 = g[:, :D].contiguous()
ms, _ = t(lambda: g1.t().mm(x)); print(f'[256xM]@[Mx256] contiguous {ms:8.2f} ms {2*M*D*D/ms/1e9:6.1f} TF')
ms, _ = t(lambda: g.sum(0)); print(f'g.sum(0)                  {ms:8.2f} ms')
ones = torch.ones(1, M, device=dev)
ms, _ = t(lambda: ones.mm(g)); print(f'ones.mm(g)                {ms:8.2f} ms')
for S in (64, 128, 256):
    def f(S=S):
        return torch.bmm(g1.view(S, M // S, D).transpose(1, 2), x.view(S, M // S, D)).sum(0)
    ms, r = t(f); print(f'256x256 bmm S={S:4d} + sum   {ms:8.2f} ms')
def f2():
    return g.view(256, M // 256, 3 * D).sum(1).sum(0)
ms, _ = t(f2); print(f'two-level sum             {ms:8.2f} ms')
