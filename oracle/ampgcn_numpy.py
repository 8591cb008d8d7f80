"""numpy restatement of the reference's AMPGCN forward + backward -- TEST INFRASTRUCTURE (oracle/__init__.py).

Follows /root/reference/src/ampnet/module/amp_gcn.py:
  :120-183  featuriser: StandardScaler z-score; tokens = cat(table[idx], z[node, idx]) (down-sampling branch,
            idx given) or cat(tile(table, feature_repeats), z[node]) for every column (full-width branch)
  :239-276  forward: conv1 -> ReLU -> conv2 -> ReLU -> token mean (or token 0) -> Linear -> log_softmax
            (dropout rates 0; dropout_adj with p = 0 is the identity)
The two AMPConv layers are oracle/ampconv_numpy.AMPConvOracle.  Parity: PINNED by tests/golden/model_*.npz,
which oracle/make_golden_ampgcn.py wrote from the reference's own class (tests/test_oracle.py).
"""
import numpy as np

from .ampconv_numpy import AMPConvOracle
from .featurizer_numpy import zscore


class AMPGCNOracle:
    def __init__(self, state, num_heads, average_pooling=True, feature_repeats=1, dtype=np.float64):
        """state: dict of the reference's state-dict arrays (keys as in AMPGCN.state_dict())."""
        self.dt = np.dtype(dtype)
        f = lambda k: np.asarray(state[k], dtype=self.dt)
        self.table = f('feature_embedding_table.weight')
        self.convs = [AMPConvOracle(f(f'{c}.multi_head_attention.in_proj_weight'),
                                    f(f'{c}.multi_head_attention.in_proj_bias'),
                                    f(f'{c}.multi_head_attention.out_proj.weight'),
                                    f(f'{c}.multi_head_attention.out_proj.bias'), num_heads, dtype=dtype)
                      for c in ('conv1', 'conv2')]
        self.Wf, self.bf = f('final_linear_out.weight'), f('final_linear_out.bias')
        self.avg, self.reps = bool(average_pooling), int(feature_repeats or 1)
        self.D = self.table.shape[1] + 1

    def tokens(self, x, idx):
        xz = zscore(x).astype(self.dt)
        N = x.shape[0]
        if idx is None:                                         # full-width branch, amp_gcn.py:170-181
            table = np.tile(self.table, (self.reps, 1))
            idx = np.tile(np.arange(x.shape[1]), (N, 1))
        else:
            table = self.table
        self._idx, self._ntab = idx, table.shape[0]
        rows = np.arange(N)[:, None]
        tok = np.concatenate([table[idx], xz[rows, idx][..., None]], axis=-1)
        return tok.reshape(N, idx.shape[1] * self.D)

    def forward(self, x, edge_index, idx=None):
        t0 = self.tokens(np.asarray(x), idx)
        self.e1, _ = self.convs[0].forward(t0, edge_index, need_weights=False)
        a1 = np.maximum(self.e1, 0)
        self.e2, _ = self.convs[1].forward(a1, edge_index, need_weights=False)
        a2 = np.maximum(self.e2, 0)
        N = a2.shape[0]
        a3 = a2.reshape(N, -1, self.D)
        self._L = a3.shape[1]
        self.pooled = a3.mean(axis=1) if self.avg else a3[:, 0]
        z = self.pooled @ self.Wf.T + self.bf
        z = z - z.max(axis=1, keepdims=True)
        self.logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
        return self.logp

    def backward(self, dlogits):
        """Gradients of sum(logits * dlogits) w.r.t. every parameter, keyed like named_parameters()."""
        g = np.asarray(dlogits, dtype=self.dt)
        dz = g - np.exp(self.logp) * g.sum(axis=1, keepdims=True)          # log_softmax
        grads = {'final_linear_out.weight': dz.T @ self.pooled, 'final_linear_out.bias': dz.sum(axis=0)}
        dpool = dz @ self.Wf
        N, L, D = dpool.shape[0], self._L, self.D
        da3 = np.zeros((N, L, D), dtype=self.dt)
        if self.avg:
            da3 += dpool[:, None, :] / L
        else:
            da3[:, 0] = dpool
        d = da3.reshape(N, L * D) * (self.e2 > 0)
        for name, conv, pre in (('conv2', self.convs[1], self.e1), ('conv1', self.convs[0], None)):
            dx, dWin, dbin, dWo, dbo = conv.backward(d)
            grads[f'{name}.multi_head_attention.in_proj_weight'] = dWin
            grads[f'{name}.multi_head_attention.in_proj_bias'] = dbin
            grads[f'{name}.multi_head_attention.out_proj.weight'] = dWo
            grads[f'{name}.multi_head_attention.out_proj.bias'] = dbo
            d = dx * (pre > 0) if pre is not None else dx
        dtok = d.reshape(N, L, D)[..., : D - 1]
        dtab = np.zeros((self._ntab, D - 1), dtype=self.dt)
        np.add.at(dtab, self._idx.reshape(-1), dtok.reshape(-1, D - 1))
        if self._ntab != self.table.shape[0]:                   # tiled table: the repeats share the rows
            dtab = dtab.reshape(self.reps, self.table.shape[0], D - 1).sum(axis=0)
        grads['feature_embedding_table.weight'] = dtab
        return grads
