"""world_size-2 gloo test of the data-parallel gradient averaging
(experiments/cora_benchmark_graphsaint_distributed.py:63-66,89-94 as intended)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ampnet_amd.distributed import GradientAllReducer, broadcast_parameters
    torch.manual_seed(100 + rank)                 # ranks start from DIFFERENT parameters
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    broadcast_parameters(model, src=0)
    reducer = GradientAllReducer(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7 + rank)   # each rank draws its own batch
    local_grads = None
    for step in range(3):
        x = torch.randn(8, 6, generator=g)
        opt.zero_grad()
        model(x).pow(2).sum().backward()
        if step == 0:
            local_grads = [p.grad.clone() for p in model.parameters()]
        reducer.allreduce()
        if step == 0:
            avg = [p.grad.clone() for p in model.parameters()]
        opt.step()
    torch.save({'params': [p.detach().clone() for p in model.parameters()],
                'local': local_grads, 'avg': avg}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_gradient_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f'r{i}.pt')) for i in range(world)]
    # parameters identical on every rank after the steps
    for a, b in zip(r[0]['params'], r[1]['params']):
        assert torch.equal(a, b)
    # averaged gradient == mean of the per-rank gradients
    for l0, l1, a0, a1 in zip(r[0]['local'], r[1]['local'], r[0]['avg'], r[1]['avg']):
        torch.testing.assert_close(a0, (l0 + l1) / 2, rtol=1e-6, atol=1e-7)
        assert torch.equal(a0, a1)
    assert not torch.equal(r[0]['local'][0], r[1]['local'][0])
