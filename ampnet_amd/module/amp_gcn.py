"""AMPGCN on MI355X: the reference's 2-layer model around AMPConv with its featuriser on the device.

Mirrors reference src/ampnet/module/amp_gcn.py:20-118 (constructor arguments, sub-module names and
therefore state-dict keys: feature_embedding_table, conv1, conv2, final_linear_out) and :239-276
(forward: dropout_adj -> featurise -> conv1 -> ReLU -> conv2 -> ReLU -> token-mean-pool -> Linear
-> log_softmax).  Out of scope and not reproduced: the matplotlib/seaborn gradient and activation
plots (:278-405) and the PCA featuriser variant (:185-237).  The per-node Python sampling loop of
:132-149 is replaced by csrc/featurizer.hip; the random stream is this library's (seeded).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..conv import AMPConv
from ..graph import _stream


class _BuildTokens(torch.autograd.Function):
    """tokens[n, l] = concat(table[idx[n, l]], zscore(x)[n, idx[n, l]]); gradient to `table` only
    (x is data; the reference's x_.requires_grad_(True) leaf is never used by an optimiser)."""

    @staticmethod
    def forward(ctx, table, x, mean, inv_std, idx):
        lib = _lib.load()
        N, Fdim = x.shape
        L, De = idx.size(1), table.size(1)
        out = torch.empty(N, L, De + 1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ampconv_feat_build(x.data_ptr(), mean.data_ptr(), inv_std.data_ptr(), idx.data_ptr(),
                                              table.data_ptr(), N, Fdim, L, De, out.data_ptr(), _stream()),
                       'ampconv_feat_build')
        ctx.save_for_backward(idx)
        ctx.dims = (N, L, De, table.size(0))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        N, L, De, Fdim = ctx.dims
        dtable = torch.empty(Fdim, De, dtype=torch.float32, device=dout.device)
        dout = dout.contiguous()
        with torch.cuda.device(dout.device):
            _lib.check(lib.ampconv_feat_table_grad(dout.data_ptr(), idx.data_ptr(), N, L, De, Fdim,
                                                   dtable.data_ptr(), _stream()), 'ampconv_feat_table_grad')
        return dtable, None, None, None, None


class FeatureTokens(nn.Module):
    """z-score + present-feature sampling + embedding concat (amp_gcn.py:120-183, downsampling branch)."""

    def __init__(self, num_node_features, feat_emb_dim, num_sampled_vectors, seed=0):
        super().__init__()
        self.feature_embedding_table = nn.Embedding(num_embeddings=num_node_features, embedding_dim=feat_emb_dim)
        self.num_sampled_vectors = num_sampled_vectors
        self._seed, self._calls = int(seed), 0

    def zscore_stats(self, x):
        lib = _lib.load()
        N, Fdim = x.shape
        mean = torch.empty(Fdim, dtype=torch.float32, device=x.device)
        inv_std = torch.empty(Fdim, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ampconv_feat_zscore_stats(x.data_ptr(), N, Fdim, mean.data_ptr(), inv_std.data_ptr(),
                                                     _stream()), 'ampconv_feat_zscore_stats')
        return mean, inv_std

    def sample(self, x, seed=None):
        lib = _lib.load()
        N, Fdim = x.shape
        L = self.num_sampled_vectors
        idx = torch.empty(N, L, dtype=torch.int32, device=x.device)
        empty = torch.zeros(1, dtype=torch.int32, device=x.device)
        if seed is None:
            self._calls += 1
            seed = (self._seed * 1000003 + self._calls) & (2 ** 64 - 1)
        with torch.cuda.device(x.device):
            _lib.check(lib.ampconv_feat_sample_present(x.data_ptr(), N, Fdim, L, seed, idx.data_ptr(),
                                                       empty.data_ptr(), _stream()), 'ampconv_feat_sample_present')
        return idx, empty

    def forward(self, x, idx=None):
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError('FeatureTokens needs float32 node features on the GPU (no CPU fallback)')
        x = x.contiguous()
        if idx is None:
            idx, empty = self.sample(x)
            if int(empty.item()):
                raise ValueError('a node has no present (non-zero) feature to sample from '
                                 '(np.random.choice raises in the reference, amp_gcn.py:135)')
        mean, inv_std = self.zscore_stats(x)
        tokens = _BuildTokens.apply(self.feature_embedding_table.weight, x, mean, inv_std, idx.contiguous())
        return tokens.view(x.size(0), -1), idx


class AMPGCN(nn.Module):
    def __init__(self, device="cuda", embedding_dim=100, num_heads=2, num_node_features=1433,
                 num_sampled_vectors=40, output_dim=7, softmax_out=True, feat_emb_dim=99, val_emb_dim=1,
                 downsample_feature_vectors=True, average_pooling_flag=True, dropout_rate=0.1,
                 dropout_adj_rate=0.1, feature_repeats=5, seed=0):
        super().__init__()
        assert embedding_dim == feat_emb_dim + val_emb_dim, \
            "Feature and value dimensions do not add up to total embedding dimension"
        if not downsample_feature_vectors or not average_pooling_flag or val_emb_dim != 1:
            raise NotImplementedError('only the down-sampling / average-pooling configuration of the '
                                      'reference harnesses is implemented')
        self.device = device
        self.emb_dim = embedding_dim
        self.num_sampled_vectors = num_sampled_vectors
        self.num_node_features = num_node_features
        self.output_dim = output_dim
        self.softmax_out = softmax_out
        self.dropout_adj_rate = dropout_adj_rate
        self.sampled_node_feat_indices = None
        self.conv1_embedding = self.conv2_embedding = None
        # same sub-module names as the reference => same state-dict keys
        self._tokens = [FeatureTokens(num_node_features, feat_emb_dim, num_sampled_vectors, seed)]
        self.feature_embedding_table = self._tokens[0].feature_embedding_table
        self.conv1 = AMPConv(embed_dim=embedding_dim, num_heads=num_heads)
        self.drop1 = nn.Dropout(p=dropout_rate)
        self.conv2 = AMPConv(embed_dim=embedding_dim, num_heads=num_heads)
        self.drop2 = nn.Dropout(p=dropout_rate)
        self.final_linear_out = nn.Linear(in_features=embedding_dim, out_features=output_dim)
        self.drop3 = nn.Dropout(p=dropout_rate)
        self.act_out = nn.Sigmoid()

    def forward(self, data, feature_indices=None):
        x, edge_index = data.x.to(self.device), data.edge_index.to(self.device)
        if self.training and self.dropout_adj_rate > 0:                       # dropout_adj (amp_gcn.py:241)
            keep = torch.rand(edge_index.size(1), device=edge_index.device) >= self.dropout_adj_rate
            edge_index = edge_index[:, keep]
        x, sampled = self._tokens[0](x, feature_indices)
        self.sampled_node_feat_indices = sampled
        x = self.drop1(x)
        x = self.conv1(x, edge_index)
        self.conv1_embedding = x
        x = F.relu(x)
        x = self.drop2(x)
        x = self.conv2(x, edge_index)
        self.conv2_embedding = x
        x = F.relu(x)
        x = self.drop3(x)
        x = x.reshape(x.shape[0], x.shape[1] // self.emb_dim, self.emb_dim).mean(dim=1)   # token average pooling
        x = self.final_linear_out(x)
        return F.log_softmax(x, dim=1) if self.softmax_out else self.act_out(x)
