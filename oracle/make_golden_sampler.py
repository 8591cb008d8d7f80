"""Generate tests/golden/sampler_*.npz from the REFERENCE's vendored GraphSAINT sampler.

TEST INFRASTRUCTURE; runs only in the build container (it needs /root/reference).  The reference keeps a
copy of PyG's sampler classes inside a plotting script, visualization/visualize_graphsaint_subgraphs.py; the
script part of that file (module level: os.chdir, a Planetoid download, networkx plots) cannot run here, so
only the CLASS DEFINITIONS -- the contiguous region from its second `import os.path as osp` to the line before
`device = torch.device(` (lines 13-203) -- are executed, unmodified, straight from the reference file.

Un-vendored third-party pieces those classes call get stand-ins whose algorithms are restated from the
libraries' published sources (torch_sparse 0.6.x, torch_geometric 2.0-2.1; neither is pinned by the
reference: setup.cfg:23-24):
  torch_sparse.SparseTensor      COO sorted by (row, col) with the edge ids as values; .coo(), .storage.value(),
                                 .random_walk(start, length) (uniform out-neighbour per step, a node without
                                 out-edges stays where it is), .saint_subgraph(node_idx) (for every node of
                                 node_idx in order, its out-edges whose target is in node_idx, relabelled)
  torch_geometric Data           attribute bag with `in`, iteration over (key, item), item assignment
  tqdm                           progress bar, unused (log=False)
The random-walk STREAM is the stand-in's own (numpy Generator): what the fixtures pin is everything the
reference's own code does with given walks -- `__getitem__` (unique + induced subgraph), `__collate__`
(attribute subsetting, node_norm / edge_norm lookup) and `__compute_norm__` (sample counting loop, the norm
formulas with their clamp / NaN / zero-count rules).  The walks are stored so that the GPU sampler can replay them.

    python oracle/make_golden_sampler.py
"""
import os
import sys
import types

import numpy as np
import torch

REF_FILE = '/root/reference/visualization/visualize_graphsaint_subgraphs.py'
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


class _Storage:
    def __init__(self, value):
        self._value = value

    def value(self):
        return self._value


class SparseTensor:
    """Stand-in for torch_sparse.SparseTensor (what GraphSAINTSampler uses of it)."""
    walk_log = None            # list that collects every random_walk result (the fixtures replay them)
    rng = None

    def __init__(self, row, col, value, sparse_sizes, is_sorted=False):
        if not is_sorted:
            order = np.lexsort((col.numpy(), row.numpy()))             # by row, then col (stable)
            order = torch.from_numpy(order)
            row, col, value = row[order], col[order], value[order]
        self.row, self.col, self.N = row, col, sparse_sizes[0]
        self.storage = _Storage(value)
        self.rowptr = torch.zeros(self.N + 1, dtype=torch.long)
        self.rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=self.N), 0)

    def coo(self):
        return self.row, self.col, self.storage.value()

    def random_walk(self, start, walk_length):
        rp, col = self.rowptr.numpy(), self.col.numpy()
        cur = start.numpy().copy()
        out = [cur.copy()]
        for _ in range(walk_length):
            deg = rp[cur + 1] - rp[cur]
            pick = (SparseTensor.rng.random(cur.shape[0]) * np.maximum(deg, 1)).astype(np.int64)
            nxt = np.where(deg > 0, col[np.minimum(rp[cur] + pick, len(col) - 1)], cur)
            cur = nxt
            out.append(cur.copy())
        walks = torch.from_numpy(np.stack(out, axis=1))
        if SparseTensor.walk_log is not None:
            SparseTensor.walk_log.append(walks.numpy().copy())
        return walks

    def saint_subgraph(self, node_idx):
        assoc = torch.full((self.N,), -1, dtype=torch.long)
        assoc[node_idx] = torch.arange(node_idx.numel())
        rows, cols, eids = [], [], []
        rp = self.rowptr
        for i, n in enumerate(node_idx.tolist()):
            a, b = int(rp[n]), int(rp[n + 1])
            c = assoc[self.col[a:b]]
            keep = c >= 0
            rows.append(torch.full((int(keep.sum()),), i, dtype=torch.long))
            cols.append(c[keep])
            eids.append(torch.arange(a, b)[keep])
        row, col, eidx = torch.cat(rows), torch.cat(cols), torch.cat(eids)
        out = SparseTensor(row=row, col=col, value=self.storage.value()[eidx],
                           sparse_sizes=(node_idx.numel(), node_idx.numel()), is_sorted=True)
        return out, eidx


class Data:
    """Stand-in for torch_geometric.data.Data (attribute bag)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, key):
        return key in self.__dict__

    def __iter__(self):
        return iter(list(self.__dict__.items()))

    def __setitem__(self, key, item):
        self.__dict__[key] = item

    @property
    def num_edges(self):
        return self.edge_index.size(1)


def load_reference_sampler():
    src = open(REF_FILE).read()
    a = src.index('import os.path as osp')
    b = src.index("device = torch.device(")
    ts = types.ModuleType('torch_sparse')
    ts.SparseTensor = SparseTensor
    tq = types.ModuleType('tqdm')
    tq.tqdm = lambda *a, **k: None
    sys.modules['torch_sparse'], sys.modules['tqdm'] = ts, tq
    ns = {'__name__': 'ref_graphsaint'}
    exec(compile(src[a:b], REF_FILE, 'exec'), ns)                  # the reference's class definitions, as they stand
    return ns['GraphSAINTRandomWalkSampler']


def run(Sampler, name, N, edge_index, batch_size, walk_length, num_steps, sample_coverage, seed):
    torch.manual_seed(seed)
    SparseTensor.rng = np.random.default_rng(seed)
    SparseTensor.walk_log = []
    E = edge_index.shape[1]
    data = Data(edge_index=torch.from_numpy(edge_index), num_nodes=N,
                x=torch.arange(N, dtype=torch.float32).view(N, 1) * 2.0, y=torch.arange(N) % 7,
                edge_attr=torch.arange(E, dtype=torch.float32) + 0.5)
    loader = Sampler(data, batch_size=batch_size, walk_length=walk_length, num_steps=num_steps,
                     sample_coverage=sample_coverage, log=False)
    norm_walks = list(SparseTensor.walk_log)                       # the samples __compute_norm__ drew
    SparseTensor.walk_log = []
    batches = list(loader)                                          # one epoch: num_steps collated batches
    out = dict(N=N, edge_index=edge_index, batch_size=batch_size, walk_length=walk_length, num_steps=num_steps,
               sample_coverage=sample_coverage, norm_walks=np.stack(norm_walks),
               node_norm=loader.node_norm.numpy(), edge_norm=loader.edge_norm.numpy(),
               epoch_walks=np.stack(SparseTensor.walk_log), n_batches=len(batches))
    for i, b in enumerate(batches):
        out[f'b{i}_edge_index'] = b.edge_index.numpy()
        out[f'b{i}_x'] = b.x.numpy()
        out[f'b{i}_y'] = b.y.numpy()
        out[f'b{i}_edge_attr'] = b.edge_attr.numpy()
        out[f'b{i}_node_norm'] = b.node_norm.numpy()
        out[f'b{i}_edge_norm'] = b.edge_norm.numpy()
        out[f'b{i}_num_nodes'] = b.num_nodes
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print(name, 'N', N, 'E', E, 'norm samples', len(norm_walks), 'batches', len(batches),
          'nodes per batch', [int(b.num_nodes) for b in batches])


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    Sampler = load_reference_sampler()
    rng = np.random.default_rng(20221007)
    # a Cora-like sparse graph with isolated nodes, a sink (no out-edge), duplicate edges and self loops
    N, E = 300, 1100
    src = rng.integers(0, N - 10, E)                                # the last 10 nodes never start an edge
    dst = rng.integers(0, N, E)
    ei = np.stack([src, dst]).astype(np.int64)
    ei[:, :20] = ei[:, 20:40]                                       # duplicates
    ei[1, 40:50] = ei[0, 40:50]                                     # self loops
    run(Sampler, 'sampler_rw', N, ei, batch_size=6, walk_length=12, num_steps=3, sample_coverage=4, seed=41)


if __name__ == '__main__':
    main()
