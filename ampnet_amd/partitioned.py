"""Destination-partitioned full-graph AMPConv across GPUs ("next" row 4 of SURVEY.md section 8f).

The reference has no counterpart (its distributed script shards independent subgraphs,
experiments/cora_benchmark_graphsaint_distributed.py:63-94; that mode is ampnet_amd/distributed.py).
This module runs ONE graph un-sampled on W ranks:

  * nodes are cut into W equal contiguous ranges; rank r holds the rows of x of its range and
    every edge whose DESTINATION lies in it (sources anywhere);
  * forward: per-node projection of the local rows, ONE all-gather of the K|V columns
    ([N*L, 2D], 2/3 of the projection), then the ordinary edge kernels over the local destination
    rows with K/V views into the gathered buffer -- the Q side, the mean and the out-projection stay
    local;
  * backward: destination pass local; the source pass produces dK|dV partial sums for EVERY source
    node from the local edges, ONE reduce-scatter (sum) returns each rank the rows of its own nodes;
    the projection GEMMs are local and the parameter gradients are SUMS over ranks
    (`GradientAllReducer(..., average=False)`).

Two collectives per layer and direction-pair, each moving 2*N*L*D elements: at BASELINE config 4
(1 M nodes, L=20, D=256, fp32) 41 GB per rank per collective -- the 7-link xGMI ring of an 8-GPU node
carries that in the time the edge kernels of a 1/8 partition need, so it wants overlapping with the
Q-side GEMMs on a second stream (not done here: correctness first).

Backend "nccl" (= RCCL) uses all_gather_into_tensor / reduce_scatter_tensor; any other backend (the
gloo test, two processes on one GPU) emulates both with all_reduce, which gloo supports on device
tensors.
"""
import os

import torch
import torch.distributed as dist

from . import _lib
from .conv import functional as F_
from .graph import EdgeCSR, _stream


class NodePartition:
    """Equal contiguous node ranges: rank r owns [r * n_local, (r+1) * n_local) of the padded range."""

    def __init__(self, num_nodes, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.num_nodes = int(num_nodes)
        self.n_local = (self.num_nodes + self.world - 1) // self.world
        self.n_padded = self.n_local * self.world
        self.begin = self.rank * self.n_local
        self.native = dist.is_initialized() and dist.get_backend(group) == 'nccl'
        # test switch: take the asynchronous RCCL branch of the two collectives even as the only rank, so that a
        # one-GPU box runs the start / wait ordering and buffer lifetimes of _PartitionedFunction for real
        self.force_async = self.native and os.environ.get('AMPCONV_PARTITION_FORCE_ASYNC') == '1'

    def local_rows(self, x_full):
        """This rank's rows of a full [N, F] tensor, zero-padded to n_local rows."""
        out = x_full.new_zeros(self.n_local, x_full.size(1))
        end = min(self.num_nodes, self.begin + self.n_local)
        if end > self.begin:
            out[: end - self.begin] = x_full[self.begin:end]
        return out

    def local_edges(self, edge_index):
        """Edges whose destination this rank owns, destination ids made local (sources stay global)."""
        dst = edge_index[1]
        keep = (dst >= self.begin) & (dst < self.begin + self.n_local)
        ei = edge_index[:, keep].clone()
        ei[1] -= self.begin
        return ei.contiguous()

    def all_gather_rows(self, t):
        """[rows, C] per rank -> [world * rows, C], rank order."""
        t = t.contiguous()
        if self.world == 1:
            return t
        out = torch.empty(self.world * t.size(0), t.size(1), dtype=t.dtype, device=t.device)
        if self.native:
            dist.all_gather_into_tensor(out, t, group=self.group)
        else:
            out.zero_()
            out[self.rank * t.size(0):(self.rank + 1) * t.size(0)] = t
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out

    # ---- the same two collectives, started asynchronously: `wait()` on the returned handle before the result is
    # used.  With RCCL the collective runs on the communicator's own stream (xGMI links + a few CUs), so the dense
    # GEMMs issued between start and wait overlap it; gloo (tests) completes inside the call.
    class _Done:
        def wait(self):
            return True

    def all_gather_rows_start(self, t):
        t = t.contiguous()
        if (self.world == 1 and not self.force_async) or not self.native:
            return self.all_gather_rows(t), self._Done()
        out = torch.empty(self.world * t.size(0), t.size(1), dtype=t.dtype, device=t.device)
        return out, dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)

    def reduce_scatter_rows_start(self, t):
        t = t.contiguous()
        if (self.world == 1 and not self.force_async) or not self.native:
            return self.reduce_scatter_rows(t), self._Done()
        out = torch.empty(t.size(0) // self.world, t.size(1), dtype=t.dtype, device=t.device)
        return out, dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_scatter_rows(self, t):
        """[world * rows, C] partial sums per rank -> [rows, C]: the sum over ranks of this rank's slice."""
        t = t.contiguous()
        if self.world == 1:
            return t
        rows = t.size(0) // self.world
        if self.native:
            out = torch.empty(rows, t.size(1), dtype=t.dtype, device=t.device)
            dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=self.group)
            return out
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t[self.rank * rows:(self.rank + 1) * rows].clone()


class _PartitionedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, w_in, b_in, w_out, b_out, csr, num_heads, part, dtype, gemm='native'):
        lib = _lib.load()
        D = w_out.size(0)
        H = int(num_heads)
        dh = D // H
        L = x_local.size(1) // D
        nl, NP = part.n_local, part.n_padded
        dev = x_local.device
        x2 = x_local.contiguous().view(nl * L, D)
        native = F_.proj_native(gemm, x_local.dtype, D)                 # libampconv's own projection kernels
        with torch.cuda.device(dev):
            # K|V of the local rows first, their all-gather in flight while the Q projection runs
            kv_loc = (F_.proj_rows(x2, F_.proj_image(w_in[D:]), b_in[D:]) if native
                      else torch.addmm(b_in[D:], x2, w_in[D:].t()))      # [nl*L, 2D]
            kv_all, work = part.all_gather_rows_start(kv_loc)           # every node's K|V, [NP*L, 2D]
            qkv = (F_.proj_rows(x2, F_.proj_image(w_in[:D]), b_in[:D]) if native
                   else torch.addmm(b_in[:D], x2, w_in[:D].t()))         # Q of the local rows, [nl*L, D]
            work.wait()
            del kv_loc
            Qv = F_._view(qkv, 0, L, dh)
            Kv, Vv = F_._view(kv_all, 0, L, dh), F_._view(kv_all, D, L, dh)
            obar = torch.empty(nl * L, D, dtype=x_local.dtype, device=dev)
            F_.edge_forward(Qv, Kv, Vv, csr, nl, L, D, H, obar, dtype=dtype)
            if native:
                y = F_.proj_rows(obar, F_.proj_image(w_out), b_out, csr.rowptr, L)
            else:
                y = torch.addmm(b_out, obar, w_out.t())
                rc = lib.ampconv_mask_rows(y.data_ptr(), csr.rowptr.data_ptr(), nl, L * D, _lib.AMPCONV_F32, _stream())
                _lib.check(rc, 'ampconv_mask_rows')
        ctx.save_for_backward(x2, w_in, w_out, qkv, kv_all, obar)
        ctx.csr, ctx.dims, ctx.part, ctx.dtype, ctx.native = csr, (nl, NP, L, D, H), part, dtype, native
        return y.view(nl, L * D)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, w_in, w_out, qkv, kv_all, obar = ctx.saved_tensors
        csr, part = ctx.csr, ctx.part
        nl, NP, L, D, H = ctx.dims
        dh = D // H
        dev = dy.device
        with torch.cuda.device(dev):
            dy2 = dy.contiguous().view(nl * L, D)
            native = ctx.native
            if native:
                dw_out, db_out = torch.empty_like(w_out), torch.empty(D, dtype=torch.float32, device=dev)
                F_.proj_wgrad(dy2, obar, dw_out, db_out, csr.rowptr, L)
                dobar = F_.proj_rows(dy2, F_.proj_image(w_out, transpose=True))
            else:
                scratch = torch.empty((1 + _lib.COLSUM_BLOCKS) * D, dtype=torch.float32, device=dev)
                rc = lib.ampconv_masked_colsum(dy2.data_ptr(), csr.rowptr.data_ptr(), nl, L, D, scratch.data_ptr(),
                                               _lib.AMPCONV_F32, _stream())
                _lib.check(rc, 'ampconv_masked_colsum')
                db_out = scratch[:D].clone()
                dw_out = F_._tn_matmul(dy2, obar)
                dobar = dy2.mm(w_out)
            Qv, dOv = F_._view(qkv, 0, L, dh), F_._view(dobar, 0, L, dh)
            Kv, Vv = F_._view(kv_all, 0, L, dh), F_._view(kv_all, D, L, dh)
            dq = torch.empty(nl * L, D, dtype=torch.float32, device=dev)
            dkv_all = torch.empty(NP * L, 2 * D, dtype=torch.float32, device=dev)     # partial sums, all sources
            dQv = F_._view(dq, 0, L, dh)
            dKv, dVv = F_._view(dkv_all, 0, L, dh), F_._view(dkv_all, D, L, dh)
            stats = spos = None
            nstat = lib.ampconv_softmax_stats_bytes(csr.num_edges, L, D, H, ctx.dtype) if F_.SOFTMAX_STATS else 0
            if nstat:
                stats = torch.empty(nstat // 4, dtype=torch.float32, device=dev)
                spos = csr.csc_positions()
            plan, nch, ws = csr.hub_args('dst', L, D, 1)
            rc = lib.ampconv_bwd_edge_dst(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(), nl, L, D, H,
                                          dQv, plan, nch, F_._ptr(ws), F_._ptr(spos), F_._ptr(stats), None,
                                          ctx.dtype, _stream())
            _lib.check(rc, 'ampconv_bwd_edge_dst')
            plan, nch, ws = csr.hub_args('src', L, D, 2)
            rc = lib.ampconv_bwd_edge_src(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(),
                                          csr.cinv.data_ptr(), NP, L, D, H, dKv, dVv, plan, nch, F_._ptr(ws),
                                          F_._ptr(stats), None, ctx.dtype, _stream())
            _lib.check(rc, 'ampconv_bwd_edge_src')
            del dobar, stats
            # this rank's rows of dK|dV, summed over ranks: in flight while the Q-side products run
            dkv, work = part.reduce_scatter_rows_start(dkv_all)
            if native:
                dw_in, db_in = torch.empty_like(w_in), torch.empty(3 * D, dtype=torch.float32, device=dev)
                F_.proj_wgrad(dq, x2, dw_in[:D], db_in[:D])
                dx = F_.proj_rows(dq, F_.proj_image(w_in[:D], transpose=True)) if ctx.needs_input_grad[0] else None
                work.wait()
                del dkv_all
                F_.proj_wgrad(dkv, x2, dw_in[D:], db_in[D:])
                if dx is not None:
                    dx = dx.add_(F_.proj_rows(dkv, F_.proj_image(w_in[D:], transpose=True))).view(nl, L * D)
            else:
                dw_q = F_._tn_matmul(dq, x2)
                db_q = dq.sum(dim=0)
                dx = dq.mm(w_in[:D]) if ctx.needs_input_grad[0] else None
                work.wait()
                del dkv_all
                dw_in = torch.cat([dw_q, F_._tn_matmul(dkv, x2)], dim=0)
                db_in = torch.cat([db_q, dkv.sum(dim=0)])
                if dx is not None:
                    dx = dx.addmm_(dkv, w_in[D:]).view(nl, L * D)
        return dx, dw_in, db_in, dw_out, db_out, None, None, None, None, None


class PartitionedAMPConv(torch.nn.Module):
    """Wraps an `ampnet_amd.AMPConv` (its parameters, replicated on every rank):

        part = NodePartition(N)                       # after dist.init_process_group
        layer = PartitionedAMPConv(AMPConv(D, H).to(dev), part)
        graph = layer.prepare(edge_index)             # once per graph: local edges -> CSR/CSC
        y_local = layer(part.local_rows(x), graph)    # [n_local, L*D]: this rank's rows of AMPConv(x, edge_index)
        loss.backward(); GradientAllReducer(layer.parameters()).allreduce(average=False)
    """

    def __init__(self, conv, partition):
        super().__init__()
        self.conv = conv
        self.partition = partition

    def prepare(self, edge_index):
        part = self.partition
        return EdgeCSR(part.local_edges(edge_index), part.n_padded)

    def forward(self, x_local, graph):
        conv, part = self.conv, self.partition
        conv._check_x(x_local, 'x_local')
        if x_local.size(0) != part.n_local:
            raise ValueError(f'x_local must hold the {part.n_local} rows of this rank (NodePartition.local_rows)')
        if x_local.dtype != torch.float32 or not conv.softmax:
            raise ValueError('the partitioned layer runs the float32 softmax path')
        return _PartitionedFunction.apply(x_local, *conv._params(), graph, conv.num_heads, part,
                                          _lib.AMPCONV_F32, conv.gemm_precision)
