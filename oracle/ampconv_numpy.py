"""Node-projected numpy restatement of AMPConv forward + explicit backward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  This is the algorithm the HIP
kernels implement, written with plain numpy so that it shares no code with
either torch autograd or the kernels.

What it follows in the reference (paths relative to /root/reference):
  * src/ampnet/conv/amp_conv.py:24-26   forward -> propagate(edge_index, x=x)
  * src/ampnet/conv/amp_conv.py:28-51   message: reshape [E, L*D] -> [E, L, D];
        MultiheadAttention(query = x_i (dst), key = value = x_j (src))
  * src/ampnet/conv/amp_conv.py:11      aggr='mean' (PyG scatter-mean; rows
        with no in-edge stay 0 -- pinned by
        synthetic_benchmark/testing_message_passing_pyg.py:37-40)
  * torch/nn/functional.py:5785-5862    packed in-projection, enc-dec branch
  * torch/nn/functional.py:6576-6612    scale, QK^T, softmax(dim=-1), PV,
        out-projection, head-mean of the weights
  * softmax=False: the reference's softmax-free attention
        (src/ampnet/conv/custom_multihead_attn_forward.py:4173-4184: q / sqrt(dh), bmm, softmax
        line commented out at :4179-4180, bmm with v), the variant amp_conv.py:6,17 refers to.
        Evaluated per edge here; the GPU path collapses it algebraically (conv/linear.py).

Re-association relative to the reference: Q depends only on the destination
node and K/V only on the source node, so they are projected once per node;
the out-projection is linear and commutes with the mean, so it is applied once
per node and masked by (in-degree > 0).
"""
import numpy as np


def _softmax_rows(s):
    m = s.max(axis=-1, keepdims=True)
    p = np.exp(s - m)
    return p / p.sum(axis=-1, keepdims=True)


class AMPConvOracle:
    """forward()/backward() for one AMPConv layer.

    Parameters use torch.nn.MultiheadAttention's names and shapes:
      in_proj_weight [3D, D], in_proj_bias [3D], out_proj_weight [D, D],
      out_proj_bias [D].
    """

    def __init__(self, in_proj_weight, in_proj_bias, out_proj_weight, out_proj_bias,
                 num_heads, dtype=np.float64, edge_chunk=4096, softmax=True):
        self.softmax = bool(softmax)
        self.dt = np.dtype(dtype)
        self.Win = np.asarray(in_proj_weight, dtype=self.dt)
        self.bin = np.asarray(in_proj_bias, dtype=self.dt)
        self.Wo = np.asarray(out_proj_weight, dtype=self.dt)
        self.bo = np.asarray(out_proj_bias, dtype=self.dt)
        self.D = self.Wo.shape[0]
        self.H = int(num_heads)
        assert self.D % self.H == 0
        self.dh = self.D // self.H
        self.edge_chunk = edge_chunk

    # ------------------------------------------------------------------ fwd
    def forward(self, x, edge_index, need_weights=True):
        D, H, dh = self.D, self.H, self.dh
        x = np.asarray(x, dtype=self.dt)
        N = x.shape[0]
        if x.shape[1] % D != 0:
            raise ValueError("invalid configuration: row width not a multiple of embed_dim")
        L = x.shape[1] // D
        src = np.asarray(edge_index[0], dtype=np.int64)
        dst = np.asarray(edge_index[1], dtype=np.int64)
        E = src.shape[0]
        X = x.reshape(N, L, D)
        Wq, Wk, Wv = self.Win[:D], self.Win[D:2 * D], self.Win[2 * D:]
        bq, bk, bv = self.bin[:D], self.bin[D:2 * D], self.bin[2 * D:]
        Q = (X @ Wq.T + bq).reshape(N, L, H, dh)
        K = (X @ Wk.T + bk).reshape(N, L, H, dh)
        V = (X @ Wv.T + bv).reshape(N, L, H, dh)
        scale = self.dt.type(1.0) / np.sqrt(self.dt.type(dh))
        deg = np.bincount(dst, minlength=N).astype(np.int64)
        Osum = np.zeros((N, L, H, dh), dtype=self.dt)
        W = np.zeros((E, L, L), dtype=self.dt) if need_weights else None
        for a in range(0, E, self.edge_chunk):
            b = min(E, a + self.edge_chunk)
            q = Q[dst[a:b]]                       # [e, L, H, dh]
            k = K[src[a:b]]
            v = V[src[a:b]]
            s = np.einsum('eihc,ejhc->ehij', q * scale, k)   # [e, H, L, L]
            p = _softmax_rows(s) if self.softmax else s
            o = np.einsum('ehij,ejhc->eihc', p, v)           # [e, L, H, dh]
            np.add.at(Osum, dst[a:b], o)
            if need_weights:
                W[a:b] = p.mean(axis=1)
        inv = 1.0 / np.maximum(deg, 1).astype(self.dt)
        Obar = Osum.reshape(N, L, D) * inv[:, None, None]
        mask = (deg > 0).astype(self.dt)
        Y = (Obar @ self.Wo.T + self.bo) * mask[:, None, None]
        self._cache = dict(X=X, Q=Q, K=K, V=V, Obar=Obar, deg=deg, mask=mask,
                           src=src, dst=dst, scale=scale, N=N, L=L, E=E)
        return Y.reshape(N, L * D), W

    # ------------------------------------------------------------------ bwd
    def backward(self, dy):
        c = self._cache
        D, H, dh = self.D, self.H, self.dh
        N, L, E = c['N'], c['L'], c['E']
        src, dst, scale = c['src'], c['dst'], c['scale']
        Q, K, V, X = c['Q'], c['K'], c['V'], c['X']
        G = np.asarray(dy, dtype=self.dt).reshape(N, L, D) * c['mask'][:, None, None]
        dWo = np.einsum('nlo,nli->oi', G, c['Obar'])
        dbo = G.sum(axis=(0, 1))
        dObar = G @ self.Wo
        inv = 1.0 / np.maximum(c['deg'], 1).astype(self.dt)
        dOn = (dObar * inv[:, None, None]).reshape(N, L, H, dh)
        dQ = np.zeros_like(Q)
        dK = np.zeros_like(K)
        dV = np.zeros_like(V)
        for a in range(0, E, self.edge_chunk):
            b = min(E, a + self.edge_chunk)
            d_, s_ = dst[a:b], src[a:b]
            q, k, v, do = Q[d_], K[s_], V[s_], dOn[d_]
            s = np.einsum('eihc,ejhc->ehij', q * scale, k)
            p = _softmax_rows(s) if self.softmax else s
            dv = np.einsum('ehij,eihc->ejhc', p, do)
            dp = np.einsum('eihc,ejhc->ehij', do, v)
            if self.softmax:
                delta = (dp * p).sum(axis=-1, keepdims=True)
                ds = p * (dp - delta)
            else:
                ds = dp
            dq = np.einsum('ehij,ejhc->eihc', ds, k) * scale
            dk = np.einsum('ehij,eihc->ejhc', ds, q) * scale
            np.add.at(dQ, d_, dq)
            np.add.at(dK, s_, dk)
            np.add.at(dV, s_, dv)
        dQ, dK, dV = (t.reshape(N, L, D) for t in (dQ, dK, dV))
        Wq, Wk, Wv = self.Win[:D], self.Win[D:2 * D], self.Win[2 * D:]
        dX = dQ @ Wq + dK @ Wk + dV @ Wv
        dWin = np.concatenate([np.einsum('nlo,nli->oi', g, X) for g in (dQ, dK, dV)], axis=0)
        dbin = np.concatenate([g.sum(axis=(0, 1)) for g in (dQ, dK, dV)], axis=0)
        return dX.reshape(N, L * D), dWin, dbin, dWo, dbo

    def attn_output(self):
        """Per-edge attention output [E, L, D] (amp_conv.py:39 `self.attn_output`)."""
        c = self._cache
        H, dh, D = self.H, self.dh, self.D
        q = c['Q'][c['dst']]
        k = c['K'][c['src']]
        v = c['V'][c['src']]
        p = np.einsum('eihc,ejhc->ehij', q * c['scale'], k)
        if self.softmax:
            p = _softmax_rows(p)
        o = np.einsum('ehij,ejhc->eihc', p, v).reshape(c['E'], c['L'], D)
        return o @ self.Wo.T + self.bo


def segment_mean(msg, index, dim_size):
    """PyG aggr='mean' (amp_conv.py:11; testing_message_passing_pyg.py:37-40)."""
    msg = np.asarray(msg)
    out = np.zeros((dim_size,) + msg.shape[1:], dtype=msg.dtype)
    np.add.at(out, index, msg)
    cnt = np.bincount(index, minlength=dim_size)
    return out / np.maximum(cnt, 1).astype(msg.dtype).reshape((-1,) + (1,) * (msg.ndim - 1))
