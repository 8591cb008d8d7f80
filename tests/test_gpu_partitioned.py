"""Destination-partitioned full-graph AMPConv (ampnet_amd/partitioned.py): two
processes (gloo; both on cuda:0 here, one per GPU over RCCL on the 8-GPU node) each own half of the
nodes; their outputs, input gradients and summed parameter gradients must equal the single-process
layer on the whole graph."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import assert_close_scaled

pytestmark = pytest.mark.gpu

N, E, L, D, H = 301, 2600, 20, 128, 4          # odd N: the last range is padded


def _problem():
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, L * D, generator=g)
    dy = torch.randn(N, L * D, generator=g)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :300] = 11                             # a hub destination inside rank 0's range
    ei[0, 300:600] = 250                         # a hub source owned by rank 1, feeding both ranges
    ei[1, ei[1] == 200] = 201                    # node 200 receives nothing
    return x, dy, ei


def _layer(dev):
    from ampnet_amd import AMPConv
    torch.manual_seed(5)
    layer = AMPConv(D, H).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    return layer


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ampnet_amd import NodePartition, PartitionedAMPConv
    from ampnet_amd.distributed import GradientAllReducer
    dev = torch.device('cuda:0')
    x, dy, ei = _problem()
    part = NodePartition(N)
    layer = PartitionedAMPConv(_layer(dev), part)
    graph = layer.prepare(ei.to(dev))
    xl = part.local_rows(x.to(dev)).requires_grad_(True)
    y = layer(xl, graph)
    y.backward(part.local_rows(dy.to(dev)))
    GradientAllReducer(layer.parameters()).allreduce(average=False)
    torch.save({'y': y.detach().cpu(), 'dx': xl.grad.cpu(),
                'grads': [p.grad.detach().cpu() for p in layer.parameters()]}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_partitioned_layer_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f'r{i}.pt')) for i in range(2)]
    dev = torch.device('cuda:0')
    x, dy, ei = _problem()
    layer = _layer(dev)
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg, ei.to(dev))
    y.backward(dy.to(dev))
    y_part = torch.cat([r[0]['y'], r[1]['y']])[:N].numpy()
    dx_part = torch.cat([r[0]['dx'], r[1]['dx']])[:N].numpy()
    assert_close_scaled(y_part, y.detach().cpu().numpy(), 'y')
    assert (y_part[200] == 0).all()
    assert_close_scaled(dx_part, xg.grad.cpu().numpy(), 'dx')
    for g0, g1, p in zip(r[0]['grads'], r[1]['grads'], layer.parameters()):
        assert torch.equal(g0, g1)                                     # summed over ranks, identical everywhere
        assert_close_scaled(g0.numpy(), p.grad.cpu().numpy(), 'parameter gradient')
    # ... and against the ORACLE on the whole graph (SURVEY.md 8f row 4 has no reference counterpart: the oracle of
    # the un-partitioned layer, pinned to the reference's fixtures by tests/test_oracle.py, is the checker)
    from oracle.ampconv_numpy import AMPConvOracle
    m = layer.multi_head_attention
    o = AMPConvOracle(*(t.detach().cpu().numpy() for t in (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight,
                                                          m.out_proj.bias)), H)
    y_ref, _ = o.forward(x.numpy(), ei.numpy())
    dx_ref, dwin, dbin, dwo, dbo = o.backward(dy.numpy())
    assert_close_scaled(y_part, y_ref, 'y vs oracle')
    assert_close_scaled(dx_part, dx_ref, 'dx vs oracle')
    for got, ref, name in zip(r[0]['grads'], (dwin, dbin, dwo, dbo), ('dW_in', 'db_in', 'dW_out', 'db_out')):
        assert_close_scaled(got.numpy(), ref, name + ' vs oracle')


def test_node_partition_bookkeeping():
    from ampnet_amd import NodePartition
    part = NodePartition(10)                      # no process group: one rank owns everything
    assert (part.world, part.rank, part.n_local, part.n_padded, part.begin) == (1, 0, 10, 10, 0)
    ei = torch.tensor([[0, 3, 9], [9, 0, 3]])
    assert torch.equal(part.local_edges(ei), ei)
    x = torch.arange(20.).view(10, 2)
    assert torch.equal(part.local_rows(x), x)


def test_rccl_collectives_single_rank():
    """The RCCL calls of the partitioned layer (`all_gather_into_tensor` / `reduce_scatter_tensor`, started with
    async_op=True as partitioned.py does) executed by one rank on the one GPU of the test box: the same code path
    as on the 8-GPU node, where only the world size differs."""
    import subprocess
    import sys
    code = '''
import os, socket, torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()      # any free port
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
t = torch.arange(12, dtype=torch.float32, device=dev).view(4, 3)
out = torch.empty(4, 3, device=dev)
w = dist.all_gather_into_tensor(out, t, async_op=True)
x = t @ t.t()                      # dense work issued between start and wait
w.wait()
assert torch.equal(out, t)
rs = torch.empty(4, 3, device=dev)
w = dist.reduce_scatter_tensor(rs, t.clone(), op=dist.ReduceOp.SUM, async_op=True)
w.wait()
assert torch.equal(rs, t)
flat = torch.ones(1000, device=dev)
dist.all_reduce(flat)
dist.broadcast(flat, src=0)
dist.barrier()
assert float(flat.sum()) == 1000.0 and x.shape == (4, 4)
dist.destroy_process_group()
print("rccl ok")
'''
    r = subprocess.run([sys.executable, '-c', code], timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and 'rccl ok' in r.stdout, r.stderr[-3000:]


def test_partitioned_layer_async_rccl_branch_single_rank():
    """the overlapped path of _PartitionedFunction (all_gather_into_tensor / reduce_scatter_tensor with
    async_op=True, wait() between the Q-side and KV-side products, `del` of the in-flight source buffers) taken for
    real: one rank under an nccl group with AMPCONV_PARTITION_FORCE_ASYNC=1, against the single-process layer."""
    import subprocess
    import sys
    from conftest import ROOT
    code = f'''
import os, socket, sys, torch, torch.distributed as dist
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                  AMPCONV_PARTITION_FORCE_ASYNC="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from conftest import assert_close_scaled
from test_gpu_partitioned import _problem, _layer, N
from ampnet_amd import NodePartition, PartitionedAMPConv
x, dy, ei = _problem()
part = NodePartition(N)
assert part.native and part.force_async and part.world == 1
ref = _layer(dev)
xg = x.to(dev).requires_grad_(True)
y = ref(xg, ei.to(dev)); y.backward(dy.to(dev))
layer = PartitionedAMPConv(_layer(dev), part)
xl = part.local_rows(x.to(dev)).requires_grad_(True)
yp = layer(xl, layer.prepare(ei.to(dev))); yp.backward(part.local_rows(dy.to(dev)))
assert_close_scaled(yp.detach().cpu().numpy(), y.detach().cpu().numpy(), "y")
assert_close_scaled(xl.grad.cpu().numpy(), xg.grad.cpu().numpy(), "dx")
for p, q in zip(layer.parameters(), ref.parameters()):
    assert_close_scaled(p.grad.cpu().numpy(), q.grad.cpu().numpy(), "parameter gradient")
dist.destroy_process_group()
print("async partitioned ok")
'''
    r = subprocess.run([sys.executable, '-c', code], timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and 'async partitioned ok' in r.stdout, r.stderr[-3000:]
