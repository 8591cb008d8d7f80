"""One AMPConv layer on ONE fixed graph as two captured HIP graphs (forward, backward).

The reference's own regime is small: Cora's full graph (2 708 nodes / 10 556 edges, `experiments/cora_benchmark_full.py`)
and GraphSAINT batches of about a thousand nodes (`experiments/cora_benchmark_graphsaint.py:77-82,96-116`).  At that size
a layer call is ~16 kernel launches of a few microseconds each and the step time is the HOST's: Python + ctypes + the
launch path, ~0.4 ms against ~0.2 ms of kernel time.  For a graph that stays the same from step to step (full-graph
training, or a validation graph) the launches of forward and of backward are recorded once and replayed:

    conv = AMPConv(128, 4).cuda()
    fast = GraphedAMPConv(conv, x_example, edge_index)      # records; x_example only gives shape / dtype / requires_grad
    y = fast(x)                                             # one graph launch
    y.backward(dy)                                          # one graph launch; conv's parameters receive .grad as usual

Same parameters (the wrapped layer's own), same kernels, same results bit for bit (tests/test_gpu_parity.py::
test_graphed_layer_matches_eager).  What is NOT captured: the graph preparation (CSR / CSC build, one launch + one status
read-back) -- it runs once, here; batches whose shape changes every step (GraphSAINT) keep the eager path.
Built on torch.cuda.make_graphed_callables (HIP graphs on ROCm): the C-ABI launches go to torch's current stream, which is
the capturing stream during recording; nothing in a small-operand layer call synchronises or reads back.
"""
import torch

from .graph import graph_cache


class _FixedGraph(torch.nn.Module):
    def __init__(self, layer, edge_index):
        super().__init__()
        self.layer = layer
        self.edge_index = edge_index

    def forward(self, x):
        return self.layer(x, self.edge_index)


class GraphedAMPConv(torch.nn.Module):
    def __init__(self, layer, x_example, edge_index, warmup=3):
        super().__init__()
        if not (x_example.is_cuda and edge_index.is_cuda):
            raise ValueError('GraphedAMPConv records HIP graphs: x and edge_index must be on the GPU')
        self.layer = layer
        self.edge_index = edge_index
        self._shape, self._dtype = tuple(x_example.shape), x_example.dtype
        # graph preparation once, outside the recording (it reads a status word back); the layer finds it by the identity
        # of `edge_index`, and this reference keeps it alive however many other graphs pass through the cache
        self._csr = graph_cache.get(edge_index, x_example.size(0))
        layer.retain_attention = False            # per-edge side outputs belong to the eager path
        sample = torch.zeros_like(x_example).requires_grad_(x_example.requires_grad)
        fixed = _FixedGraph(layer, edge_index)
        self._pin = _PinnedCache(edge_index, x_example.size(0), self._csr)
        with self._pin:
            self._call = torch.cuda.make_graphed_callables(fixed, (sample,), num_warmup_iters=warmup)

    def forward(self, x):
        if tuple(x.shape) != self._shape or x.dtype != self._dtype:
            raise ValueError(f'recorded for x of shape {self._shape} / {self._dtype}, got {tuple(x.shape)} / {x.dtype}')
        return self._call(x)


class _PinnedCache:
    """While recording, the layer's graph_cache.get must return the prepared graph (no build, no read-back)."""

    def __init__(self, edge_index, n, csr):
        self.edge_index, self.n, self.csr = edge_index, n, csr

    def __enter__(self):
        self._orig = graph_cache.get

        def get(edge_index, num_nodes):
            if edge_index is self.edge_index and int(num_nodes) == int(self.n):
                return self.csr
            return self._orig(edge_index, num_nodes)
        graph_cache.get = get
        return self

    def __exit__(self, *exc):
        graph_cache.get = self._orig
        return False
