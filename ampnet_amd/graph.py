"""Edge-list -> CSR/CSC on device, through the C ABI (csrc/graph_prep.hip).

PyG's MessagePassing.propagate consumes the unsorted `edge_index` directly
(reference src/ampnet/conv/amp_conv.py:25); the HIP edge kernels instead walk a
destination-sorted CSR (forward, dQ) and a source-sorted CSC (dK, dV), both
built here with stable sorts.
"""
import weakref

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class EdgeCSR:
    """dst-sorted CSR (rowptr, col, eperm) + src-sorted CSC (cscptr, crow, cperm), int32,
    and cinv[p] = 1 / in-degree(crow[p]) (fp32)."""

    def __init__(self, edge_index, num_nodes, validate=True):
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f'edge_index must have shape [2, E], got {tuple(edge_index.shape)}')
        if edge_index.dtype != torch.int64:
            raise ValueError(f'edge_index must be int64, got {edge_index.dtype}')
        if not edge_index.is_cuda:
            raise ValueError('edge_index must live on the GPU (ampnet_amd has no CPU path)')
        lib = _lib.load()
        ei = edge_index.contiguous()
        dev = ei.device
        E, N = int(ei.size(1)), int(num_nodes)
        self.num_nodes, self.num_edges, self.device = N, E, dev
        i32 = dict(dtype=torch.int32, device=dev)
        self.rowptr = torch.empty(N + 1, **i32)
        self.cscptr = torch.empty(N + 1, **i32)
        self.col = torch.empty(max(E, 1), **i32)
        self.eperm = torch.empty(max(E, 1), **i32)
        self.crow = torch.empty(max(E, 1), **i32)
        self.cperm = torch.empty(max(E, 1), **i32)
        self.cinv = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
        oob = torch.zeros(1, **i32)
        with torch.cuda.device(dev):
            ws_bytes = lib.ampconv_csr_workspace_bytes(N, E) if E > 0 else 0
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
            rc = lib.ampconv_csr_build(ei.data_ptr(), E, N, self.rowptr.data_ptr(), self.col.data_ptr(),
                                       self.eperm.data_ptr(), self.cscptr.data_ptr(), self.crow.data_ptr(),
                                       self.cperm.data_ptr(), self.cinv.data_ptr(), oob.data_ptr(), ws.data_ptr(),
                                       ws_bytes,
                                       _stream())
        _lib.check(rc, 'ampconv_csr_build')
        if validate and E > 0 and int(oob.item()) != 0:
            raise ValueError(f'edge_index contains node ids outside [0, {N})')

    @classmethod
    def identity(cls, n, device):
        """n rows, edge r: r -> r (used by AMPConv.message on pre-gathered pairs)."""
        self = cls.__new__(cls)
        ar = torch.arange(n + 1, dtype=torch.int32, device=device)
        self.num_nodes, self.num_edges, self.device = n, n, device
        self.rowptr = self.cscptr = ar
        self.col = self.crow = self.eperm = self.cperm = ar[:max(n, 1)]
        self.cinv = torch.ones(max(n, 1), dtype=torch.float32, device=device)
        return self


class _Cache:
    """Tiny cache so that conv1/conv2/backward of one batch share one CSR build
    (src/ampnet/module/amp_gcn.py:248,259 pass the same edge_index twice)."""

    def __init__(self, size=4):
        self.size, self.items = size, []

    def get(self, edge_index, num_nodes):
        key = (id(edge_index), edge_index._version, int(num_nodes), edge_index.data_ptr())
        for k, ref, csr in self.items:
            if k == key and ref() is edge_index:
                return csr
        csr = EdgeCSR(edge_index, num_nodes)
        self.items.append((key, weakref.ref(edge_index), csr))
        if len(self.items) > self.size:
            self.items.pop(0)
        return csr

    def clear(self):
        self.items = []


graph_cache = _Cache()
