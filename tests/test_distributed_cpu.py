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


def _partition_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ampnet_amd.partitioned import NodePartition
    from ampnet_amd.distributed import GradientAllReducer
    part = NodePartition(7)                                   # 7 nodes on 2 ranks: ranges [0,4) and [4,8) padded
    x = torch.arange(14.).view(7, 2)
    ei = torch.tensor([[0, 1, 2, 3, 4, 5, 6, 6], [6, 5, 4, 3, 2, 1, 0, 6]])
    gathered = part.all_gather_rows(part.local_rows(x))       # emulated with all_reduce on gloo
    partial = torch.full((part.n_padded, 3), float(rank + 1))
    scattered = part.reduce_scatter_rows(partial)
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.full((3,), float(rank + 1))
    GradientAllReducer([p]).allreduce(average=False)
    torch.save({'n_local': part.n_local, 'begin': part.begin, 'edges': part.local_edges(ei),
                'gathered': gathered, 'scattered': scattered, 'grad': p.grad.clone()},
               os.path.join(out_dir, f'p{rank}.pt'))
    dist.destroy_process_group()


def test_node_partition_world2(tmp_path):
    """Bookkeeping and collectives of the destination-partitioned layer (ampnet_amd/partitioned.py) on
    two gloo ranks: equal padded ranges, edges filed by destination with local ids, gather /
    reduce-scatter of row blocks, SUMMED parameter gradients."""
    mp.spawn(_partition_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f'p{i}.pt')) for i in range(2)]
    assert [r[0]['n_local'], r[0]['begin'], r[1]['begin']] == [4, 0, 4]
    assert torch.equal(r[0]['edges'], torch.tensor([[3, 4, 5, 6], [3, 2, 1, 0]]))          # dst 0..3 stay
    assert torch.equal(r[1]['edges'], torch.tensor([[0, 1, 2, 6], [2, 1, 0, 2]]))          # dst 4..6 -> 0..2
    want = torch.cat([torch.arange(14.).view(7, 2), torch.zeros(1, 2)])
    assert torch.equal(r[0]['gathered'], want) and torch.equal(r[1]['gathered'], want)
    assert torch.equal(r[0]['scattered'], torch.full((4, 3), 3.0)) and torch.equal(r[1]['scattered'], torch.full((4, 3), 3.0))
    assert torch.equal(r[0]['grad'], torch.full((3,), 3.0)) and torch.equal(r[1]['grad'], torch.full((3,), 3.0))
