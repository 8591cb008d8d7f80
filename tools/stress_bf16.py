"""Developer stress test: the bf16 kernels that read MFMA operands with ds_read_b64_tr_b16 (weight-gradient product, edge
passes) launched many times on the same inputs must give the same bits every time ("a bug found on the way")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ampnet_amd import AMPConv  # noqa: E402
from ampnet_amd.conv import functional as F_  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
# ---- weight-gradient product
for (M, Na, Nb, L) in ((60000, 128, 128, 20), (60000, 768, 256, 20), (60000, 256, 256, 20)):
    a = torch.randn(M, Na, device=dev).bfloat16()
    b = torch.randn(M, Nb, device=dev).bfloat16()
    deg = (torch.rand(M // L, device=dev) < 0.8).int()
    rp = torch.zeros(M // L + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    for masked in (True, False):
        dw0 = torch.empty(Na, Nb, device=dev, dtype=torch.bfloat16)
        cs0 = torch.empty(Na, device=dev, dtype=torch.bfloat16)
        F_.proj_wgrad(a, b, dw0, cs0, rp if masked else None, L)
        bad = 0
        for it in range(300):
            dw, cs = torch.empty_like(dw0), torch.empty_like(cs0)
            if it % 2:
                junk = torch.randn(1 << 20, device=dev)
            F_.proj_wgrad(a, b, dw, cs, rp if masked else None, L)
            bad += int(not (torch.equal(dw, dw0) and torch.equal(cs, cs0)))
        print('wgrad bf16', M, Na, Nb, 'masked' if masked else 'plain', 'mismatching runs:', bad, '/ 300', flush=True)
# ---- the layer (edge passes): outputs and gradients of repeated steps
import bench  # noqa: E402
for (N, E, L, D, H) in ((4000, 60000, 20, 256, 8), (3000, 40000, 20, 128, 8), (1 << 15, 600000, 20, 256, 8)):
    layer = AMPConv(D, H).to(dev).to(torch.bfloat16)
    x0 = torch.randn(N, L * D, device=dev).bfloat16()
    dy = torch.randn(N, L * D, device=dev).bfloat16()
    if N & (N - 1) == 0:            # R-MAT (config 5's generator): id-structured degrees, long segments
        ei = bench.rmat_edges(N.bit_length() - 1, E, torch.Generator(device=dev).manual_seed(7), dev)
    else:
        ei = torch.randint(0, N, (2, E), device=dev)
        ei[1, : E // 10] = 3
        ei[0, E // 10: E // 5] = 7
    ref = None
    bad = 0
    steps = 600 if N < 10000 else 200
    for it in range(steps):
        layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        y = layer(x, ei)
        y.backward(dy)
        m = layer.multi_head_attention
        out = [y.detach(), x.grad] + [p.grad.clone() for p in (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias)]
        if ref is None:
            ref = out
        else:
            bad += int(not all(torch.equal(u, v) for u, v in zip(out, ref)))
    print('layer bf16', N, E, L, D, H, 'mismatching steps:', bad, '/', steps - 1, flush=True)
