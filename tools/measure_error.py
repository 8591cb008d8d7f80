"""Dev: max error of the GPU layer against the fp64 numpy oracle, relative to the largest entry of each
tensor (2 000 nodes / 24 000 edges, L=20, D=256, H=8), for every projection mode side by side.

    python tools/measure_error.py [--gemm native|fp32|bf16x3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ampnet_amd import AMPConv  # noqa: E402
from oracle.ampconv_numpy import AMPConvOracle  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    N, E, L, D, H = 2000, 24000, 20, 256, 8
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, L * D, generator=g)
    dy = torch.randn(N, L * D, generator=g)
    ei = torch.randint(0, N, (2, E), generator=g)
    torch.manual_seed(1)
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    m = layer.multi_head_attention
    o = AMPConvOracle(*(t.detach().cpu().double().numpy() for t in (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight,
                                                                     m.out_proj.bias)), H)
    y_ref, _ = o.forward(x.double().numpy(), ei.numpy(), need_weights=False)
    ref = (y_ref,) + tuple(o.backward(dy.double().numpy()))
    names = ('y', 'dx', 'd in_proj_weight', 'd in_proj_bias', 'd out_proj.weight', 'd out_proj.bias')
    # 'native' = the scaled two-plane projections (forced on: this graph is below their size threshold),
    # 'native-6' = the six-product form
    from ampnet_amd.conv import functional as F_
    modes = [sys.argv[sys.argv.index('--gemm') + 1]] if '--gemm' in sys.argv else ['native', 'native-6', 'fp32', 'bf16x3']
    res = {}
    for mode in modes:
        layer.gemm_precision = mode.split('-')[0]
        F_.PROJ_SCALED_MIN_ELEMENTS = 0 if mode == 'native' else 1 << 62
        layer.zero_grad(set_to_none=True)
        xg = x.to(dev).requires_grad_(True)
        y = layer(xg, ei.to(dev))
        y.backward(dy.to(dev))
        got = (y.detach(), xg.grad, m.in_proj_weight.grad, m.in_proj_bias.grad, m.out_proj.weight.grad, m.out_proj.bias.grad)
        res[mode] = [float(np.abs(a.double().cpu().numpy() - b).max() / np.abs(b).max()) for a, b in zip(got, ref)]
    print(f'N={N} E={E} L={L} D={D} H={H}; max |error| / max |reference entry| against the fp64 oracle')
    print('| quantity | ' + ' | '.join(modes) + ' |')
    print('|---|' + '---|' * len(modes))
    for i, name in enumerate(names):
        print(f'| {name} | ' + ' | '.join(f'{res[mode][i]:.1e}' for mode in modes) + ' |')


if __name__ == '__main__':
    main()
