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
    """Raw hipStream_t of torch's current stream on the current device (the C-level accessors: the Stream object of
    torch.cuda.current_stream() costs ~12 us per call, five calls per Cora-sized step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


# node lists are worth their extra launches when at most this share of the nodes has any edge
ACTIVE_LIST_FRACTION = 0.85


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
        status = torch.empty(4, **i32)                   # {bounds flag, dst chunks, src chunks, -}: ONE read-back
        # CSC position of every original edge id (what csc_positions() derives `spos` from): a by-product of the
        # one-launch preparation of small graphs (GraphSAINT batches, Cora: saves two launches per step there); big
        # graphs build it lazily, and only if a pass asks for the statistics hand-off (bf16 storage never does:
        # 160 MB and a scatter launch per build at BASELINE config 5)
        self._by_edge = torch.empty(max(E, 1), **i32) if E <= 12288 else None
        with torch.cuda.device(dev):
            ws_bytes = lib.ampconv_csr_workspace_bytes(N, E) if E > 0 else 0
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
            # long-segment plans (hubs of power-law graphs): one for the dst-sorted CSR, one for the
            # src-sorted CSC, made by the same call
            self.hub_dst = self.hub_src = None
            self.hub_dst_chunks = self.hub_src_chunks = 0
            plans = [None, None]
            self.hub_chunk = chunk = _lib.hub_chunk(E)
            if E > chunk:
                nb = lib.ampconv_hub_plan_bytes(E, chunk)
                plans = [torch.empty(nb // 4, **i32), torch.empty(nb // 4, **i32)]
            rc = lib.ampconv_graph_build(ei.data_ptr(), E, N, self.rowptr.data_ptr(), self.col.data_ptr(),
                                         self.eperm.data_ptr(), self.cscptr.data_ptr(), self.crow.data_ptr(),
                                         self.cperm.data_ptr(), self.cinv.data_ptr(), status.data_ptr(), chunk,
                                         plans[0].data_ptr() if plans[0] is not None else None,
                                         plans[1].data_ptr() if plans[1] is not None else None,
                                         self._by_edge.data_ptr() if self._by_edge is not None else None,
                                         ws.data_ptr(), ws_bytes, _stream())
            _lib.check(rc, 'ampconv_graph_build')
            bad = 0
            if plans[0] is not None or (validate and E > 0):
                bad, self.hub_dst_chunks, self.hub_src_chunks, _ = status.tolist()
                if self.hub_dst_chunks:
                    self.hub_dst = plans[0][: 4 + 4 * self.hub_dst_chunks]
                if self.hub_src_chunks:
                    self.hub_src = plans[1][: 4 + 4 * self.hub_src_chunks]
        if validate and E > 0 and bad != 0:
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
        self.hub_dst = self.hub_src = None
        self.hub_dst_chunks = self.hub_src_chunks = 0
        self.hub_chunk = 0
        return self

    def csc_positions(self):
        """spos[p] = CSC position of the edge at CSR position p (lazily, once per graph): where the
        destination pass files the softmax statistics the source pass reads back in order."""
        if getattr(self, '_spos', None) is None:
            lib = _lib.load()
            E = self.num_edges
            if self.eperm.data_ptr() == self.cperm.data_ptr():        # identity graph
                self._spos = self.eperm
            else:
                self._spos = torch.empty(max(E, 1), dtype=torch.int32, device=self.device)
                with torch.cuda.device(self.device):
                    if self._by_edge is not None:
                        rc = lib.ampconv_csc_positions_from(self.eperm.data_ptr(), self._by_edge.data_ptr(), E,
                                                            self._spos.data_ptr(), _stream())
                        _lib.check(rc, 'ampconv_csc_positions_from')
                    else:
                        scratch = torch.empty(max(E, 1), dtype=torch.int32, device=self.device)
                        rc = lib.ampconv_csc_positions(self.eperm.data_ptr(), self.cperm.data_ptr(), E,
                                                       scratch.data_ptr(), self._spos.data_ptr(), _stream())
                        _lib.check(rc, 'ampconv_csc_positions')
                self._by_edge = None                                  # derived once per graph: not kept
        return self._spos

    def active_nodes(self):
        """{'in' | 'out' | 'any': (ids int32 [count + 8], count, ptr int32 [N + 1])} -- the nodes with at least one
        in-edge / out-edge / either, for the projections' node lists (include/ampconv.h: ampconv_active_nodes), or None
        when (nearly) every node has edges.  Built once per graph: three launches of a flag-scan-fill and ONE read-back.
        The per-node formulation projects every node; the reference's per-edge one (amp_conv.py:36-47) never touches a
        node that no edge names."""
        if not hasattr(self, '_active'):
            lib = _lib.load()
            N = self.num_nodes
            self._active = None
            if N > 0 and self.num_edges > 0 and self.rowptr.data_ptr() != self.cscptr.data_ptr():
                i32 = dict(dtype=torch.int32, device=self.device)
                counts = torch.empty(3, **i32)
                out = {}
                with torch.cuda.device(self.device):
                    nws = lib.ampconv_active_nodes_workspace_bytes(N)
                    ws = torch.empty(nws, dtype=torch.uint8, device=self.device)
                    for i, (name, which) in enumerate((('in', 1), ('out', 2), ('any', 3))):
                        ids, ptr = torch.empty(N + 8, **i32), torch.empty(N + 1, **i32)
                        rc = lib.ampconv_active_nodes(self.rowptr.data_ptr(), self.cscptr.data_ptr(), N, which,
                                                      ids.data_ptr(), ptr.data_ptr(), counts[i:].data_ptr(),
                                                      ws.data_ptr(), nws, _stream())
                        _lib.check(rc, 'ampconv_active_nodes')
                        out[name] = (ids, ptr)
                    cnt = counts.tolist()
                if cnt[2] <= ACTIVE_LIST_FRACTION * N:
                    self._active = {name: (out[name][0][:c + 8], c, out[name][1])
                                    for name, c in zip(('in', 'out', 'any'), cnt)}
        return self._active

    def hub_args(self, side, L, D, n_tiles):
        """(plan pointer, n_chunks, workspace tensor) of the 'dst' or 'src' long-segment plan."""
        plan = self.hub_dst if side == 'dst' else self.hub_src
        n = self.hub_dst_chunks if side == 'dst' else self.hub_src_chunks
        if plan is None or n == 0:
            return None, 0, None
        ws = torch.empty(n * L * D * n_tiles, dtype=torch.float32, device=plan.device)
        return plan.data_ptr(), n, ws


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
