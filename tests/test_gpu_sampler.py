"""GraphSAINT random-walk sampler on the GPU against the numpy restatement of the reference's
vendored sampler (oracle/graphsaint_numpy.py).  The random stream is unpinned (torch_sparse is
absent); its defining properties and everything downstream of it are checked."""
import types

import numpy as np
import pytest
import torch

from oracle import graphsaint_numpy as ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def graph():
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5)
    N, E = 3000, 24000
    ei = rng.integers(0, N - 30, size=(2, E)).astype(np.int64)       # last 30 nodes isolated
    ei[0, :40] = N - 1                                                # ... except node N-1: out-edges only
    data = types.SimpleNamespace(x=torch.randn(N, 12, device=dev), y=torch.randint(0, 7, (N,), device=dev),
                                 edge_index=torch.from_numpy(ei).to(dev),
                                 train_mask=(torch.rand(N, device=dev) < 0.5),
                                 edge_attr=torch.arange(E, device=dev, dtype=torch.float32))
    return data, ei, N, E


def test_walks_and_induced_subgraph(graph):
    from ampnet_amd.sampler import GraphSAINTRandomWalkSampler
    data, ei, N, E = graph
    s = GraphSAINTRandomWalkSampler(data, batch_size=20, walk_length=30, num_steps=3, seed=11, num_nodes=N)
    seen = []
    for _ in range(3):
        node_idx, edge_index, edge_id, walks = s.sample()
        w = walks.cpu().numpy()
        assert w.shape == (20, 31)
        assert ref.walk_is_valid(ei, N, w)
        n_ref, e_ref, keep = ref.induced_subgraph(ei, N, w)
        np.testing.assert_array_equal(node_idx.cpu().numpy(), n_ref)          # sorted unique nodes
        got = edge_index.cpu().numpy()
        eid = edge_id.cpu().numpy()
        order = np.argsort(eid)                                               # edge order is free
        np.testing.assert_array_equal(eid[order], keep)
        np.testing.assert_array_equal(got[:, order], e_ref)
        # relabelled edges point at the right original endpoints
        np.testing.assert_array_equal(n_ref[got[0]], ei[0][eid])
        np.testing.assert_array_equal(n_ref[got[1]], ei[1][eid])
        seen.append(w.copy())
    assert not np.array_equal(seen[0], seen[1])                               # a new draw each call
    s2 = GraphSAINTRandomWalkSampler(data, batch_size=20, walk_length=30, num_steps=3, seed=11, num_nodes=N)
    np.testing.assert_array_equal(s2.sample()[3].cpu().numpy(), seen[0])      # same seed -> same batch


def test_walk_steps_are_uniform_over_neighbours(graph):
    from ampnet_amd.sampler import GraphSAINTRandomWalkSampler
    data, ei, N, E = graph
    s = GraphSAINTRandomWalkSampler(data, batch_size=4000, walk_length=1, seed=3, num_nodes=N)
    # all walks start at node N-1 (40 out-edges): first steps must be ~uniform over its neighbours
    walks = torch.empty(4000, 2, dtype=torch.int64, device=data.x.device)
    from ampnet_amd import _lib
    start = torch.full((4000,), N - 1, dtype=torch.int64, device=data.x.device)
    _lib.check(_lib.load().ampconv_saint_random_walk(s.csr.cscptr.data_ptr(), s.csr.crow.data_ptr(),
                                                     start.data_ptr(), 4000, 1, 12345, walks.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), 'walk')
    nxt = walks[:, 1].cpu().numpy()
    neigh = ei[1][ei[0] == N - 1]
    assert set(nxt.tolist()) <= set(neigh.tolist())
    counts = np.array([(nxt == v).sum() for v in np.unique(neigh)])
    mult = np.array([(neigh == v).sum() for v in np.unique(neigh)])          # duplicate edges count twice
    expected = 4000 * mult / mult.sum()
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert chi2 < 2.5 * len(counts), chi2                                      # loose goodness of fit


def test_norms_and_batches(graph):
    from ampnet_amd.sampler import GraphSAINTRandomWalkSampler
    data, ei, N, E = graph
    s = GraphSAINTRandomWalkSampler(data, batch_size=30, walk_length=20, num_steps=4, sample_coverage=3,
                                    seed=7, num_nodes=N)
    nn_ref, en_ref = ref.norms(s.node_count.cpu().numpy(), s.edge_count.cpu().numpy(), ei[0], N, s.num_samples)
    np.testing.assert_allclose(s.node_norm.cpu().numpy(), nn_ref, rtol=1e-6)
    np.testing.assert_allclose(s.edge_norm.cpu().numpy(), en_ref, rtol=1e-6)
    assert s.node_count.sum().item() >= N * 3                                 # coverage reached
    n_batches = 0
    for b in s:
        n_batches += 1
        idx = b.node_idx
        assert torch.equal(b.x, data.x[idx]) and torch.equal(b.y, data.y[idx])
        assert torch.equal(b.train_mask, data.train_mask[idx])
        assert torch.equal(b.edge_attr, data.edge_attr[b.edge_id])            # per-edge attributes follow
        assert torch.equal(b.node_norm, s.node_norm[idx]) and torch.equal(b.edge_norm, s.edge_norm[b.edge_id])
        assert b.edge_index.max().item() < b.num_nodes
    assert n_batches == len(s) == 4


def test_sampler_feeds_the_layer(graph):
    """The harness pattern of experiments/cora_benchmark_graphsaint.py:96-110 on the GPU path."""
    from ampnet_amd import AMPConv
    from ampnet_amd.sampler import GraphSAINTRandomWalkSampler
    data, ei, N, E = graph
    dev = data.x.device
    feats = types.SimpleNamespace(x=torch.randn(N, 4 * 32, device=dev), y=data.y, edge_index=data.edge_index)
    layer = AMPConv(32, 1).to(dev)
    s = GraphSAINTRandomWalkSampler(feats, batch_size=8, walk_length=50, num_steps=2, sample_coverage=1,
                                    seed=1, num_nodes=N)
    for b in s:
        out = layer(b.x, b.edge_index)
        loss = (out.pow(2).mean(dim=1) * b.node_norm).sum()
        loss.backward()
    assert torch.isfinite(layer.multi_head_attention.in_proj_weight.grad).all()


def test_replay_of_reference_sampler_fixture():
    """tests/golden/sampler_rw.npz was produced by the reference's own vendored GraphSAINT classes
    (visualization/visualize_graphsaint_subgraphs.py:13-203, executed unmodified by oracle/make_golden_sampler.py over
    stand-ins for torch_sparse / PyG Data).  Replaying its walks through the GPU sampler must give the same node sets,
    induced edges, subset attributes and -- from the same sampling sequence -- the same node_norm / edge_norm."""
    import os
    from conftest import GOLDEN_DIR, load_golden
    from ampnet_amd.sampler import GraphSAINTRandomWalkSampler
    g = load_golden(os.path.join(GOLDEN_DIR, 'sampler_rw.npz'))
    dev = torch.device('cuda:0')
    ei, N = g['edge_index'], int(g['N'])
    E = ei.shape[1]
    data = types.SimpleNamespace(edge_index=torch.from_numpy(ei).to(dev), num_nodes=N,
                                 x=(torch.arange(N, dtype=torch.float32).view(N, 1) * 2.0).to(dev),
                                 y=(torch.arange(N) % 7).to(dev),
                                 edge_attr=(torch.arange(E, dtype=torch.float32) + 0.5).to(dev))
    s = GraphSAINTRandomWalkSampler(data, batch_size=int(g['batch_size']), walk_length=int(g['walk_length']),
                                    num_steps=int(g['num_steps']), sample_coverage=0, seed=1, num_nodes=N)
    s.sample_coverage = int(g['sample_coverage'])
    s.node_norm, s.edge_norm = s._compute_norm(walks=[torch.from_numpy(w) for w in g['norm_walks']])
    assert s.num_samples == len(g['norm_walks'])                              # the reference's stopping rule
    np.testing.assert_allclose(s.node_norm.cpu().numpy(), g['node_norm'], rtol=1e-6)
    np.testing.assert_allclose(s.edge_norm.cpu().numpy(), g['edge_norm'], rtol=1e-6)
    for i, w in enumerate(g['epoch_walks']):
        node_idx, edge_index, edge_id, _ = s.sample(walks=torch.from_numpy(w))
        b = s._collate(node_idx, edge_index, edge_id)
        assert b.num_nodes == int(g[f'b{i}_num_nodes'])
        np.testing.assert_array_equal(b.x.cpu().numpy(), g[f'b{i}_x'])
        np.testing.assert_array_equal(b.y.cpu().numpy(), g[f'b{i}_y'])
        np.testing.assert_allclose(b.node_norm.cpu().numpy(), g[f'b{i}_node_norm'], rtol=1e-6)
        want = sorted(zip(g[f'b{i}_edge_index'][0].tolist(), g[f'b{i}_edge_index'][1].tolist(),
                          g[f'b{i}_edge_attr'].tolist(), np.round(g[f'b{i}_edge_norm'], 5).tolist()))
        got = sorted(zip(b.edge_index[0].tolist(), b.edge_index[1].tolist(), b.edge_attr.tolist(),
                         np.round(b.edge_norm.cpu().numpy(), 5).tolist()))
        assert got == want


def test_replayed_walks_are_range_checked():
    """sample(walks=...) feeds the ids into mark[node] = 1 on the device; ids outside [0, N) must be a
    Python error, not a device write out of bounds."""
    import types
    from ampnet_amd import GraphSAINTRandomWalkSampler
    dev = torch.device('cuda:0')
    N = 50
    ei = torch.randint(0, N, (2, 400), device=dev)
    s = GraphSAINTRandomWalkSampler(types.SimpleNamespace(edge_index=ei, num_nodes=N), batch_size=4, walk_length=5,
                                    num_nodes=N)
    good = torch.randint(0, N, (4, 6), device=dev)
    node_idx, sub_ei, edge_id, _ = s.sample(walks=good)
    assert node_idx.numel() == good.unique().numel() and int(sub_ei.max()) < node_idx.numel()
    for bad in (good.clone().index_fill_(0, torch.tensor([0], device=dev), N), good - N, good.flatten()):
        with pytest.raises(ValueError):
            s.sample(walks=bad)


def test_gather_rows_matches_indexing():
    """The collate step's row gather (csrc/sampler.hip, ampconv_saint_gather_rows) against torch indexing: fp32 and bf16
    rows, a strided view, repeated and out-of-order indices; tensors it does not serve fall through to torch."""
    from ampnet_amd.sampler import gather_rows
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(2)
    for shape, dtype in (((500, 20 * 256), torch.float32), ((300, 40 * 100), torch.float32), ((257, 20, 256), torch.bfloat16),
                         ((64, 4), torch.float32), ((90, 5000), torch.float32)):
        x = torch.randn(*shape, generator=g).to(dtype).to(dev)
        idx = torch.randint(0, shape[0], (777,), generator=g).to(dev)
        assert torch.equal(gather_rows(x, idx), x[idx]), (shape, dtype)
    x = torch.randn(400, 1024, generator=g).to(dev)
    view = x[:, :512]                                   # rows 4096 bytes apart, 2048 bytes long
    idx = torch.arange(399, -1, -1, device=dev)
    assert torch.equal(gather_rows(view, idx), view[idx])
    y = torch.randint(0, 7, (400,), generator=g).to(dev)      # labels: 1-D, torch's path
    assert torch.equal(gather_rows(y, idx), y[idx])
    assert gather_rows(x, idx[:0]).shape == (0, 1024)
