"""numpy/sklearn restatement of the AMPGCN featuriser -- TEST INFRASTRUCTURE (oracle/__init__.py).

Follows /root/reference/src/ampnet/module/amp_gcn.py:
  :122-125  StandardScaler().fit_transform(x)   (sklearn, the reference's own dependency)
  :146-147  tokens = cat(embedding_table[idx], x_[node, idx][:, None])
  :152-153  reshape to [N, L * D]
The sampling of :132-135 (np.random.choice over the present features, with replacement) is random:
UNPINNED stream; `indices_are_present` states its defining property.
"""
import numpy as np
from sklearn.preprocessing import StandardScaler


def zscore(x):
    return StandardScaler().fit_transform(np.asarray(x)).astype(np.float32)


def build_tokens(x, idx, table):
    xz = zscore(x)
    N, L = idx.shape
    rows = np.arange(N)[:, None]
    tok = np.concatenate([table[idx], xz[rows, idx][..., None]], axis=-1)
    return tok.reshape(N, L * (table.shape[1] + 1)).astype(np.float32)


def indices_are_present(x, idx):
    rows = np.arange(idx.shape[0])[:, None]
    return bool((np.asarray(x)[rows, idx] != 0).all())
