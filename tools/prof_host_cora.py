"""Developer tool: where the HOST spends a Cora-sized step (2 708 nodes / 10 556 edges, L=20, D=128, H=4) -- cProfile over
200 steps of forward + backward with a fresh graph per step (the GraphSAINT regime: every batch is a new graph).

    python tools/prof_host_cora.py [--fixed-graph] [--graphed]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ampnet_amd import AMPConv, graph_cache  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    N, E, L, D, H = 2708, 10556, 20, 128, 4
    layer = AMPConv(D, H).to(dev)
    layer.retain_attention = False
    x = torch.randn(N, L * D, device=dev, requires_grad=True)
    dy = torch.randn(N, L * D, device=dev)
    ei = torch.randint(0, N, (2, E), device=dev)
    fixed = '--fixed-graph' in sys.argv

    def step():
        if not fixed:
            graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        x.grad = None
        y = layer(x, ei)
        y.backward(dy)

    if '--graphed' in sys.argv:                 # forward and backward replayed as HIP graphs (ampnet_amd.GraphedAMPConv)
        from ampnet_amd import GraphedAMPConv
        fast = GraphedAMPConv(layer, x, ei)

        def step():                             # noqa: F811
            layer.zero_grad(set_to_none=True)
            x.grad = None
            fast(x).backward(dy)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    print(f'{(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per step (fixed graph: {fixed})')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)


if __name__ == '__main__':
    main()
