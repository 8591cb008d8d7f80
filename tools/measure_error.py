"""Dev: max error of the GPU layer against the fp64 numpy oracle, relative to the largest entry of each
tensor (2 000 nodes / 24 000 edges, L=20, D=256, H=8), for both GEMM modes."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from ampnet_amd import AMPConv
from oracle.ampconv_numpy import AMPConvOracle
dev = torch.device('cuda:0')
N, E, L, D, H = 2000, 24000, 20, 256, 8
g = torch.Generator().manual_seed(5)
x = torch.randn(N, L * D, generator=g); dy = torch.randn(N, L * D, generator=g)
ei = torch.randint(0, N, (2, E), generator=g)
torch.manual_seed(1)
layer = AMPConv(D, H).to(dev)
with torch.no_grad():
    layer.multi_head_attention.in_proj_bias.normal_(0, 0.1); layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
m = layer.multi_head_attention
o = AMPConvOracle(*(t.detach().cpu().numpy() for t in (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias)), H)
y_ref, _ = o.forward(x.numpy(), ei.numpy(), need_weights=False)
ref = o.backward(dy.numpy())
for gemm in ('fp32', 'bf16x3'):
    layer.gemm_precision = gemm
    layer.zero_grad(set_to_none=True)
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg, ei.to(dev)); y.backward(dy.to(dev))
    def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
    print(gemm, 'y %.2e dx %.2e dWin %.2e dbin %.2e dWo %.2e' % (
        rel(y.detach().cpu().numpy(), y_ref), rel(xg.grad.cpu().numpy(), ref[0]), rel(m.in_proj_weight.grad.cpu().numpy(), ref[1]),
        rel(m.in_proj_bias.grad.cpu().numpy(), ref[2]), rel(m.out_proj.weight.grad.cpu().numpy(), ref[3])))
