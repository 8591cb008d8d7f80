"""Softmax-free AMPConv ("next" row 3 of SURVEY.md section 8f) on MI355X.

The reference carries a copy of nn.MultiheadAttention whose scaled-dot-product step has the softmax
line removed (reference src/ampnet/conv/custom_multihead_attn_forward.py:4173-4184, softmax
commented out at :4179-4180; class at custom_multihead_attn.py:13); amp_conv.py:6,17 refer to it as
"Custom multihead attention removing softmax layer".  Per edge e = (s -> d) and head h it computes

    O_e = (Q_d K_s^T / sqrt(dh)) V_s

Without the softmax the product re-associates: mean_e O_e = Q_d (mean_e K_s^T V_s) / sqrt(dh), so the
L x L scores never exist here.  Per SOURCE node and head one dh x dh matrix M_s = K_s^T V_s
(`ampconv_linear_outer`), then the whole edge phase is a segment mean of M rows over the dst-sorted CSR
(`ampconv_gather_segment_sum`, HBM-bound: D*dh floats per edge instead of 2*L*D), then Q_d Mbar_d per
destination (`ampconv_linear_apply`); all three are HIP kernels of libampconv.so.  Backward is the transpose: dMbar_d = Q_d^T dObar_d,
dM_s = sum over the out-edges of s of dMbar_d / deg_d (same kernel over the CSC with the per-edge
1/deg weights), dK_s = V_s dM_s^T, dV_s = K_s dM_s, dQ_d = dObar_d Mbar_d^T.
"""
import math

import torch

from .. import _lib
from ..graph import _stream
from .functional import _tn_matmul, _view, gemm_precision


def gather_segment_sum(rows, ptr, idx, n_out, weights=None, mean=False):
    """out[r] = scale_r * sum_{p in segment r} w_p * rows[idx[p]] (include/ampconv.h)."""
    lib = _lib.load()
    rows = rows.contiguous()
    F = rows.size(1)
    out = torch.empty(n_out, F, dtype=torch.float32, device=rows.device)
    with torch.cuda.device(rows.device):
        rc = lib.ampconv_gather_segment_sum(rows.data_ptr(), ptr.data_ptr(), idx.data_ptr(),
                                            weights.data_ptr() if weights is not None else None,
                                            1 if mean else 0, n_out, F, out.data_ptr(), _stream())
    _lib.check(rc, 'ampconv_gather_segment_sum')
    return out


def _outer(A, B, n, L, D, H, scale, dev):
    """M[n, h] = scale * A[n,:,h]^T B[n,:,h] as an [n, H*dh*dh] matrix."""
    lib = _lib.load()
    dh = D // H
    M = torch.empty(n, H * dh * dh, dtype=torch.float32, device=dev)
    _lib.check(lib.ampconv_linear_outer(A, B, n, L, D, H, scale, M.data_ptr(), _stream()), 'ampconv_linear_outer')
    return M


def _apply(A, M, transpose, n, L, D, H, scale, Out):
    lib = _lib.load()
    _lib.check(lib.ampconv_linear_apply(A, M.data_ptr(), 1 if transpose else 0, n, L, D, H, scale, Out, _stream()),
               'ampconv_linear_apply')


class LinearAMPConvFunction(torch.autograd.Function):
    """y = mask(deg>0) * (Q_d mean_e(K_s^T V_s) / sqrt(dh) Wo^T + bo); `xq` supplies the destination
    rows and `xkv` the source rows (the same tensor in AMPConv.forward)."""

    @staticmethod
    def forward(ctx, xq, xkv, w_in, b_in, w_out, b_out, csr, num_heads, shared, gemm='fp32'):
        lib = _lib.load()
        D = w_out.size(0)
        H = int(num_heads)
        dh = D // H
        if (dh * dh) % 4:
            raise ValueError('the softmax-free path needs an even head dimension')
        L = xq.size(1) // D
        Nq, Nk = xq.size(0), xkv.size(0)
        dev = xq.device
        rs = 1.0 / math.sqrt(dh)
        xq2 = xq.contiguous().view(Nq * L, D)
        with torch.cuda.device(dev), gemm_precision(gemm):
            if shared:
                qkv = torch.addmm(b_in, xq2, w_in.t())
                xkv2, kv = xq2, None
                Qv, Kv, Vv = (_view(qkv, i * D, L, dh) for i in range(3))
            else:
                xkv2 = xkv.contiguous().view(Nk * L, D)
                qkv = torch.addmm(b_in[:D], xq2, w_in[:D].t())
                kv = torch.addmm(b_in[D:], xkv2, w_in[D:].t())
                Qv, Kv, Vv = _view(qkv, 0, L, dh), _view(kv, 0, L, dh), _view(kv, D, L, dh)
            M = _outer(Kv, Vv, Nk, L, D, H, 1.0, dev)                                  # K_s^T V_s
            Mbar = gather_segment_sum(M, csr.rowptr, csr.col, Nq, mean=True)           # the edge phase
            del M
            obar = torch.empty(Nq * L, D, dtype=torch.float32, device=dev)
            _apply(Qv, Mbar, False, Nq, L, D, H, rs, _view(obar, 0, L, dh))            # Q_d Mbar_d / sqrt(dh)
            y = torch.addmm(b_out, obar, w_out.t())
            rc = lib.ampconv_mask_rows(y.data_ptr(), csr.rowptr.data_ptr(), Nq, L * D, _lib.AMPCONV_F32, _stream())
            _lib.check(rc, 'ampconv_mask_rows')
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(xq2, xkv2, w_in, w_out, qkv, kv, Mbar)       # obar is recomputed (one cheap pass)
        ctx.csr, ctx.dims, ctx.shared, ctx.gemm = csr, (Nq, Nk, L, D, H), shared, gemm
        ctx.mark_non_differentiable(qkv)
        if kv is not None:
            ctx.mark_non_differentiable(kv)
        return y.view(Nq, L * D), qkv, kv

    @staticmethod
    def backward(ctx, dy, _dqkv=None, _dkv=None):
        lib = _lib.load()
        if dy is None:
            return (None,) * 10
        xq2, xkv2, w_in, w_out, qkv, kv, Mbar = ctx.saved_tensors
        csr, shared = ctx.csr, ctx.shared
        Nq, Nk, L, D, H = ctx.dims
        dh = D // H
        dev = dy.device
        rs = 1.0 / math.sqrt(dh)
        need_xq, need_xkv = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        with torch.cuda.device(dev), gemm_precision(ctx.gemm):
            dy2 = dy.contiguous().view(Nq * L, D)
            scratch = torch.empty((1 + _lib.COLSUM_BLOCKS) * D, dtype=torch.float32, device=dev)
            rc = lib.ampconv_masked_colsum(dy2.data_ptr(), csr.rowptr.data_ptr(), Nq, L, D, scratch.data_ptr(),
                                           _lib.AMPCONV_F32, _stream())
            _lib.check(rc, 'ampconv_masked_colsum')
            db_out = scratch[:D].clone()
            if shared:
                Qv, Kv, Vv = (_view(qkv, i * D, L, dh) for i in range(3))
            else:
                Qv, Kv, Vv = _view(qkv, 0, L, dh), _view(kv, 0, L, dh), _view(kv, D, L, dh)
            obar = torch.empty(Nq * L, D, dtype=torch.float32, device=dev)
            _apply(Qv, Mbar, False, Nq, L, D, H, rs, _view(obar, 0, L, dh))
            dw_out = _tn_matmul(dy2, obar)
            del obar
            # rows with no in-edge: obar = 0 and Mbar = 0, but dobar must not leak into dQ / dMbar
            dobar = dy2.mm(w_out)
            rc = lib.ampconv_mask_rows(dobar.data_ptr(), csr.rowptr.data_ptr(), Nq, L * D, _lib.AMPCONV_F32, _stream())
            _lib.check(rc, 'ampconv_mask_rows')
            dOv = _view(dobar, 0, L, dh)
            dMbar = _outer(Qv, dOv, Nq, L, D, H, rs, dev)                               # Q^T dObar / sqrt(dh)
            # transpose of the segment mean: every out-edge of s brings dMbar[dst] / deg(dst)
            dM = gather_segment_sum(dMbar, csr.cscptr, csr.crow, Nk, weights=csr.cinv)
            del dMbar
            # gradients land in the packed buffers the projection GEMMs read
            if shared:
                dqkv = torch.empty(Nq * L, 3 * D, dtype=torch.float32, device=dev)
                dQv, dKv, dVv = (_view(dqkv, i * D, L, dh) for i in range(3))
            else:
                dq2 = torch.empty(Nq * L, D, dtype=torch.float32, device=dev)
                dkv = torch.empty(Nk * L, 2 * D, dtype=torch.float32, device=dev)
                dQv, dKv, dVv = _view(dq2, 0, L, dh), _view(dkv, 0, L, dh), _view(dkv, D, L, dh)
            _apply(dOv, Mbar, True, Nq, L, D, H, rs, dQv)                               # dQ = dObar Mbar^T / sqrt(dh)
            del dobar
            _apply(Vv, dM, True, Nk, L, D, H, 1.0, dKv)                                 # dK = V dM^T
            _apply(Kv, dM, False, Nk, L, D, H, 1.0, dVv)                                # dV = K dM
            del dM
            if shared:
                dw_in = _tn_matmul(dqkv, xq2)
                db_in = dqkv.sum(dim=0)
                dxq = dqkv.mm(w_in).view(Nq, L * D) if need_xq else None
                dxkv = None
            else:
                dw_in = torch.cat([_tn_matmul(dq2, xq2), _tn_matmul(dkv, xkv2)], dim=0)
                db_in = torch.cat([dq2.sum(dim=0), dkv.sum(dim=0)])
                dxq = dq2.mm(w_in[:D]).view(Nq, L * D) if need_xq else None
                dxkv = dkv.mm(w_in[D:]).view(Nk, L * D) if need_xkv else None
        return dxq, dxkv, dw_in, db_in, dw_out, db_out, None, None, None, None


def attention_scores(Qv, Kv, edge_index, L, D, H):
    """[E, L, L] head-averaged raw scaled scores, original edge order
    (custom_multihead_attn_forward.py:4173-4175 and :4441-4442)."""
    lib = _lib.load()
    E = edge_index.size(1)
    W = torch.empty(E, L, L, dtype=torch.float32, device=edge_index.device)
    with torch.cuda.device(edge_index.device):
        rc = lib.ampconv_attn_scores(Qv, Kv, edge_index.data_ptr(), E, L, D, H, W.data_ptr(),
                                     _lib.AMPCONV_F32, _stream())
    _lib.check(rc, 'ampconv_attn_scores')
    return W
