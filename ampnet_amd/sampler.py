"""GraphSAINT random-walk sampler on the GPU (the loader in front of the hot path).

Mirrors the interface the reference's harness uses (experiments/cora_benchmark_graphsaint.py:80-82):

    loader = GraphSAINTRandomWalkSampler(data, batch_size=8, walk_length=150,
                                         num_steps=200, sample_coverage=100)
    for batch in loader:  batch.x, batch.y, batch.edge_index, batch.node_norm, batch.train_mask ...

`data` is any object with tensor attributes (x, y, edge_index, masks ...), e.g. a PyG `Data` or
`types.SimpleNamespace`; attributes whose first dimension is num_nodes / num_edges are subset
like PyG's `__collate__` does (reference copy: visualization/visualize_graphsaint_subgraphs.py
:112-135).  Everything runs in libampconv.so (csrc/sampler.hip); the graph stays on the device.
The random stream is this library's own (counter-based, seeded): torch_sparse's is not available.
"""
import types

import torch

from . import _lib
from .graph import EdgeCSR, _stream


def gather_rows(item, node_idx):
    """item[node_idx] for a resident per-node tensor (the sampler's collate step, visualize_graphsaint_subgraphs.py:112-135):
    rows of a multiple of 16 bytes on the GPU go through the library's gather kernel, anything else through torch's indexing
    (labels, masks: a few bytes per node)."""
    rb = item[0].numel() * item.element_size() if item.size(0) else 0
    if not (item.is_cuda and node_idx.is_cuda and node_idx.dtype == torch.int64 and item.dim() >= 2 and rb and rb % 16 == 0
            and item[0].is_contiguous() and (item.stride(0) * item.element_size()) % 16 == 0 and item.data_ptr() % 16 == 0
            and not item.requires_grad):
        return item[node_idx]
    out = torch.empty((node_idx.numel(),) + tuple(item.shape[1:]), dtype=item.dtype, device=item.device)
    with torch.cuda.device(item.device):
        _lib.check(_lib.load().ampconv_saint_gather_rows(item.data_ptr(), item.stride(0) * item.element_size(), rb,
                                                         node_idx.contiguous().data_ptr(), node_idx.numel(), out.data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream), 'ampconv_saint_gather_rows')
    return out


class GraphSAINTRandomWalkSampler:
    def __init__(self, data, batch_size, walk_length, num_steps=1, sample_coverage=0, seed=0,
                 num_nodes=None):
        self.data = data
        ei = data.edge_index
        if not ei.is_cuda:
            raise ValueError('the sampler runs on the GPU: move data.edge_index to the device first')
        self.N = int(num_nodes if num_nodes is not None else getattr(data, 'num_nodes', None) or data.x.size(0))
        self.E = int(ei.size(1))
        self.batch_size, self.walk_length = int(batch_size), int(walk_length)
        self.num_steps, self.sample_coverage = int(num_steps), int(sample_coverage)
        self.device = ei.device
        self.csr = EdgeCSR(ei, self.N)                         # src-sorted CSC = out-neighbour lists
        self._lib = _lib.load()
        self._gen = torch.Generator(device=self.device).manual_seed(int(seed))
        self._seed, self._draw = int(seed), 0
        i32 = dict(dtype=torch.int32, device=self.device)
        self._mark = torch.empty(self.N, **i32)
        self._relabel = torch.empty(self.N, **i32)
        self._node_buf = torch.empty(self.N, dtype=torch.int64, device=self.device)
        self._ws = torch.empty(self._lib.ampconv_saint_workspace_bytes(self.N), dtype=torch.uint8,
                               device=self.device)
        self._cnt2 = torch.zeros(2, **i32)
        self.node_norm = self.edge_norm = None
        if self.sample_coverage > 0:
            self.node_norm, self.edge_norm = self._compute_norm()

    def __len__(self):
        return self.num_steps

    # -- visualize_graphsaint_subgraphs.py:195-199 + :107-110
    def sample(self, walks=None):
        """One sub-graph: (node_idx [n_sub] sorted, edge_index [2, e_sub] relabelled, edge_id [e_sub], walks).
        `walks` ([n_walks, walk_length + 1] int64 on the device): replay given walks instead of drawing new ones
        (tests/test_gpu_sampler.py replays the walks of the reference-generated fixture)."""
        lib, dev, st = self._lib, self.device, _stream
        csr = self.csr
        with torch.cuda.device(dev):
            if walks is not None:
                walks = walks.to(device=dev, dtype=torch.int64).contiguous()
                # replayed ids go straight into mark[node] = 1 on the device: refuse what would write out of bounds
                # (one read-back, on this test / replay path only)
                if walks.dim() != 2 or walks.numel() == 0:
                    raise ValueError(f'walks must be [n_walks, walk_length + 1], got {tuple(walks.shape)}')
                lo, hi = int(walks.min()), int(walks.max())
                if lo < 0 or hi >= self.N:
                    raise ValueError(f'walks contain node ids outside [0, {self.N}): min {lo}, max {hi}')
            else:
                start = torch.randint(0, self.N, (self.batch_size,), generator=self._gen, device=dev)
                walks = torch.empty(self.batch_size, self.walk_length + 1, dtype=torch.int64, device=dev)
                self._draw += 1
                _lib.check(lib.ampconv_saint_random_walk(csr.cscptr.data_ptr(), csr.crow.data_ptr(), start.data_ptr(),
                                                         self.batch_size, self.walk_length,
                                                         (self._seed * 1000003 + self._draw) & (2 ** 64 - 1),
                                                         walks.data_ptr(), st()), 'ampconv_saint_random_walk')
            _lib.check(lib.ampconv_saint_nodes(walks.data_ptr(), walks.numel(), self.N, self._mark.data_ptr(),
                                               self._relabel.data_ptr(), self._node_buf.data_ptr(),
                                               self._cnt2.data_ptr(), self._ws.data_ptr(), self._ws.numel(), st()),
                       'ampconv_saint_nodes')
            # the number of sampled nodes stays on the device until the edge count is known too: an upper bound
            # (walked nodes, at most N) sizes the per-node counters, and both sizes come back in ONE read
            n_bound = min(int(walks.numel()), self.N)
            cnt = torch.empty(n_bound + 1, dtype=torch.int32, device=dev)
            off = torch.empty(n_bound + 1, dtype=torch.int32, device=dev)
            _lib.check(lib.ampconv_saint_count_edges_bounded(self._node_buf.data_ptr(), n_bound, self._cnt2.data_ptr(),
                                                             csr.cscptr.data_ptr(), csr.crow.data_ptr(),
                                                             self._mark.data_ptr(), cnt.data_ptr(), off.data_ptr(),
                                                             self._cnt2[1:].data_ptr(), self._ws.data_ptr(),
                                                             self._ws.numel(), st()), 'ampconv_saint_count_edges_bounded')
            n_sub, e_sub = self._cnt2[:2].tolist()
            node_idx = self._node_buf[:n_sub].clone()
            edge_index = torch.empty(2, e_sub, dtype=torch.int64, device=dev)
            edge_id = torch.empty(e_sub, dtype=torch.int64, device=dev)
            _lib.check(lib.ampconv_saint_fill_edges(node_idx.data_ptr(), n_sub, csr.cscptr.data_ptr(),
                                                    csr.crow.data_ptr(), csr.cperm.data_ptr(), self._mark.data_ptr(),
                                                    self._relabel.data_ptr(), off.data_ptr(), e_sub,
                                                    edge_index.data_ptr(), edge_id.data_ptr(), st()),
                       'ampconv_saint_fill_edges')
        return node_idx, edge_index, edge_id, walks

    # -- visualize_graphsaint_subgraphs.py:112-135
    def _collate(self, node_idx, edge_index, edge_id):
        out = types.SimpleNamespace()
        out.num_nodes = int(node_idx.numel())
        out.edge_index = edge_index
        out.node_idx, out.edge_id = node_idx, edge_id
        items = self.data.items() if hasattr(self.data, 'items') and callable(self.data.items) else vars(self.data).items()
        for key, item in items:
            if key in ('edge_index', 'num_nodes'):
                continue
            if isinstance(item, torch.Tensor) and item.dim() > 0 and item.size(0) == self.N:
                setattr(out, key, gather_rows(item, node_idx.to(item.device)))
            elif isinstance(item, torch.Tensor) and item.dim() > 0 and item.size(0) == self.E:
                setattr(out, key, item[edge_id.to(item.device)])
            else:
                setattr(out, key, item)
        if self.sample_coverage > 0:
            out.node_norm = self.node_norm[node_idx]
            out.edge_norm = self.edge_norm[edge_id]
        return out

    def __iter__(self):
        for _ in range(self.num_steps):
            node_idx, edge_index, edge_id, _ = self.sample()
            yield self._collate(node_idx, edge_index, edge_id)

    # -- visualize_graphsaint_subgraphs.py:137-173
    def _compute_norm(self, walks=None):
        """node_norm / edge_norm from repeated sampling (visualize_graphsaint_subgraphs.py:137-173).
        `walks`: an iterable of walk tensors to replay, in drawing order (tests)."""
        lib, dev = self._lib, self.device
        replay = iter(walks) if walks is not None else None
        node_count = torch.zeros(self.N, dtype=torch.float32, device=dev)
        edge_count = torch.zeros(max(self.E, 1), dtype=torch.float32, device=dev)
        num_samples = total = 0
        with torch.cuda.device(dev):
            while total < self.N * self.sample_coverage:
                for _ in range(self.num_steps):
                    node_idx, _, edge_id, _ = self.sample(next(replay) if replay is not None else None)
                    _lib.check(lib.ampconv_saint_add_counts(node_idx.data_ptr(), node_idx.numel(),
                                                            node_count.data_ptr(), _stream()), 'add_counts')
                    _lib.check(lib.ampconv_saint_add_counts(edge_id.data_ptr(), edge_id.numel(),
                                                            edge_count.data_ptr(), _stream()), 'add_counts')
                    total += int(node_idx.numel())
                num_samples += self.num_steps
            node_norm = torch.empty(self.N, dtype=torch.float32, device=dev)
            edge_norm = torch.empty(max(self.E, 1), dtype=torch.float32, device=dev)
            src = self.data.edge_index[0].contiguous()
            _lib.check(lib.ampconv_saint_norms(node_count.data_ptr(), edge_count.data_ptr(), src.data_ptr(),
                                               self.N, self.E, float(num_samples), node_norm.data_ptr(),
                                               edge_norm.data_ptr(), _stream()), 'ampconv_saint_norms')
        self.node_count, self.edge_count, self.num_samples = node_count, edge_count[:self.E], num_samples
        return node_norm, edge_norm[:self.E]
