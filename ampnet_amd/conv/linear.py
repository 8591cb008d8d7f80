"""Softmax-free AMPConv ("next" row 3 of SURVEY.md section 8f) on MI355X.

The reference carries a copy of nn.MultiheadAttention whose scaled-dot-product step has the softmax
line removed (reference src/ampnet/conv/custom_multihead_attn_forward.py:4173-4184, softmax
commented out at :4179-4180; class at custom_multihead_attn.py:13); amp_conv.py:6,17 refer to it as
"Custom multihead attention removing softmax layer".  Per edge e = (s -> d) and head h it computes

    O_e = (Q_d K_s^T / sqrt(dh)) V_s

Without the softmax the product re-associates: mean_e O_e = Q_d (mean_e K_s^T V_s) / sqrt(dh), so the
L x L scores never exist here.  Per SOURCE node and head one dh x dh matrix M_s = K_s^T V_s (a batched
GEMM over the L tokens), then the whole edge phase is a segment mean of M rows over the dst-sorted CSR
(`ampconv_gather_segment_sum`, HIP, HBM-bound: D*dh floats per edge instead of 2*L*D), then one
batched GEMM Q_d Mbar_d per destination.  Backward is the transpose: dMbar_d = Q_d^T dObar_d,
dM_s = sum over the out-edges of s of dMbar_d / deg_d (same kernel over the CSC with the per-edge
1/deg weights), dK_s = V_s dM_s^T, dV_s = K_s dM_s, dQ_d = dObar_d Mbar_d^T.
"""
import math

import torch

from .. import _lib
from ..graph import _stream
from .functional import _tn_matmul, gemm_precision


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


class LinearAMPConvFunction(torch.autograd.Function):
    """y = mask(deg>0) * (Q_d mean_e(K_s^T V_s) / sqrt(dh) Wo^T + bo); `xq` supplies the destination
    rows and `xkv` the source rows (the same tensor in AMPConv.forward)."""

    @staticmethod
    def forward(ctx, xq, xkv, w_in, b_in, w_out, b_out, csr, num_heads, shared, gemm='fp32'):
        lib = _lib.load()
        D = w_out.size(0)
        H = int(num_heads)
        dh = D // H
        if dh % 2:
            raise ValueError('the softmax-free path needs an even head dimension')
        L = xq.size(1) // D
        Nq, Nk = xq.size(0), xkv.size(0)
        xq2 = xq.contiguous().view(Nq * L, D)
        with torch.cuda.device(xq.device), gemm_precision(gemm):
            if shared:
                qkv = torch.addmm(b_in, xq2, w_in.t())
                xkv2, kv = xq2, None
                q4 = qkv[:, :D].view(Nq, L, H, dh)
                k4, v4 = qkv[:, D:2 * D].view(Nk, L, H, dh), qkv[:, 2 * D:].view(Nk, L, H, dh)
            else:
                xkv2 = xkv.contiguous().view(Nk * L, D)
                qkv = torch.addmm(b_in[:D], xq2, w_in[:D].t())
                kv = torch.addmm(b_in[D:], xkv2, w_in[D:].t())
                q4 = qkv.view(Nq, L, H, dh)
                k4, v4 = kv[:, :D].view(Nk, L, H, dh), kv[:, D:].view(Nk, L, H, dh)
            M = torch.einsum('nlhi,nlhj->nhij', k4, v4).reshape(Nk, H * dh * dh)       # K_s^T V_s
            Mbar = gather_segment_sum(M, csr.rowptr, csr.col, Nq, mean=True)           # the edge phase
            del M
            obar = (torch.einsum('nlhi,nhij->nlhj', q4, Mbar.view(Nq, H, dh, dh)) / math.sqrt(dh)).reshape(Nq * L, D)
            y = torch.addmm(b_out, obar, w_out.t())
            rc = lib.ampconv_mask_rows(y.data_ptr(), csr.rowptr.data_ptr(), Nq, L * D, _lib.AMPCONV_F32, _stream())
            _lib.check(rc, 'ampconv_mask_rows')
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(xq2, xkv2, w_in, w_out, qkv, kv, obar, Mbar)
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
        xq2, xkv2, w_in, w_out, qkv, kv, obar, Mbar = ctx.saved_tensors
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
            dw_out = _tn_matmul(dy2, obar)
            # rows with no in-edge: obar = 0 and Mbar = 0, but dobar must not leak into dQ / dMbar
            dobar = dy2.mm(w_out)
            rc = lib.ampconv_mask_rows(dobar.data_ptr(), csr.rowptr.data_ptr(), Nq, L * D, _lib.AMPCONV_F32, _stream())
            _lib.check(rc, 'ampconv_mask_rows')
            do4 = dobar.view(Nq, L, H, dh)
            if shared:
                q4 = qkv[:, :D].view(Nq, L, H, dh)
                k4, v4 = qkv[:, D:2 * D].view(Nk, L, H, dh), qkv[:, 2 * D:].view(Nk, L, H, dh)
            else:
                q4 = qkv.view(Nq, L, H, dh)
                k4, v4 = kv[:, :D].view(Nk, L, H, dh), kv[:, D:].view(Nk, L, H, dh)
            dq4 = torch.einsum('nlhj,nhij->nlhi', do4, Mbar.view(Nq, H, dh, dh)) * rs
            dMbar = (torch.einsum('nlhi,nlhj->nhij', q4, do4) * rs).reshape(Nq, H * dh * dh)
            # transpose of the segment mean: every out-edge of s brings dMbar[dst] / deg(dst)
            dM = gather_segment_sum(dMbar, csr.cscptr, csr.crow, Nk, weights=csr.cinv).view(Nk, H, dh, dh)
            del dMbar
            dk4 = torch.einsum('nlhj,nhij->nlhi', v4, dM)
            dv4 = torch.einsum('nlhi,nhij->nlhj', k4, dM)
            if shared:
                dqkv = torch.cat([dq4.reshape(Nq * L, D), dk4.reshape(Nk * L, D), dv4.reshape(Nk * L, D)], dim=1)
                dw_in = _tn_matmul(dqkv, xq2)
                db_in = dqkv.sum(dim=0)
                dxq = dqkv.mm(w_in).view(Nq, L * D) if need_xq else None
                dxkv = None
            else:
                dq2 = dq4.reshape(Nq * L, D)
                dkv = torch.cat([dk4.reshape(Nk * L, D), dv4.reshape(Nk * L, D)], dim=1)
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
