"""AMPGCN on MI355X: the reference's 2-layer model around AMPConv with its featuriser on the device.

Mirrors reference src/ampnet/module/amp_gcn.py:20-118 (constructor arguments, sub-module names and
therefore state-dict keys: feature_embedding_table, conv1, conv2, final_linear_out) and :239-276
(forward: dropout_adj -> featurise -> conv1 -> ReLU -> conv2 -> ReLU -> token pooling -> Linear
-> log_softmax).  Both featuriser branches of :120-183 are here: down-sampling of the present features
(:127-153, the Cora harness) and the full-width branch (:170-181, `downsample_feature_vectors=False`,
the XOR harness of synthetic_benchmark/xor_training_utils.py:58-72); both poolings of :268-271 (token mean,
or token 0 with `average_pooling_flag=False`).  Out of scope and not reproduced: the matplotlib/seaborn
gradient and activation plots (:278-405) and the PCA featuriser variant (:185-237).  The per-node Python
sampling loop of :132-149 is replaced by csrc/featurizer.hip; the random stream is this library's (seeded);
`forward(data, feature_indices=...)` takes the indices of another stream (tests: the reference's own).
Pinned against the reference's class by tests/golden/model_*.npz (oracle/make_golden_ampgcn.py).
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

    def forward_all(self, x, feature_repeats):
        """Full-width branch (amp_gcn.py:170-181): token f of a node = concat(tile(table, feature_repeats)[f],
        zscore(x)[node, f]) for EVERY feature column f; no sampling (returns indices None like the reference)."""
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError('FeatureTokens needs float32 node features on the GPU (no CPU fallback)')
        x = x.contiguous()
        reps = int(feature_repeats) if feature_repeats else 1
        table = self.feature_embedding_table.weight
        if reps > 1:
            table = table.repeat(reps, 1)                       # torch.tile(weight, [feature_repeats, 1])
        if table.size(0) != x.size(1) or x.size(1) != self.num_sampled_vectors:
            raise ValueError(f'full-width tokens need num_node_features * feature_repeats ({table.size(0)}) == '
                             f'x.shape[1] ({x.size(1)}) == num_sampled_vectors ({self.num_sampled_vectors}) '
                             '(torch.cat / reshape of amp_gcn.py:174-181 raise otherwise)')
        idx = torch.arange(x.size(1), dtype=torch.int32, device=x.device).repeat(x.size(0), 1)
        mean, inv_std = self.zscore_stats(x)
        tokens = _BuildTokens.apply(table.contiguous(), x, mean, inv_std, idx)
        return tokens.view(x.size(0), -1), None

    def forward(self, x, idx=None):
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError('FeatureTokens needs float32 node features on the GPU (no CPU fallback)')
        x = x.contiguous()
        if idx is not None:
            idx = torch.as_tensor(idx, device=x.device).to(torch.int32)
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
        if val_emb_dim != 1:
            raise ValueError('val_emb_dim must be 1: the reference concatenates ONE z-scored value per token '
                             '(amp_gcn.py:147,175; its reshape raises for any other value)')
        self.device = device
        self.downsampling_vectors = downsample_feature_vectors
        self.average_pooling_flag = average_pooling_flag
        self.feature_repeats = feature_repeats
        self.feat_emb_dim, self.val_emb_dim = feat_emb_dim, val_emb_dim
        self.dropout_rate = dropout_rate
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
        if not average_pooling_flag:               # defined (and in the state dict) but never used by forward,
            self.cls_token = nn.Parameter(torch.zeros(1, 1, self.emb_dim))    # exactly as amp_gcn.py:55-57,268-271
            nn.init.normal_(self.cls_token, std=.02)
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
        if self.downsampling_vectors:
            x, sampled = self._tokens[0](x, feature_indices)
        else:
            x, sampled = self._tokens[0].forward_all(x, self.feature_repeats)
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
        x = x.reshape(x.shape[0], x.shape[1] // self.emb_dim, self.emb_dim)
        x = x.mean(dim=1) if self.average_pooling_flag else x[:, 0]      # token mean / token 0 (amp_gcn.py:268-271)
        x = self.final_linear_out(x)
        return F.log_softmax(x, dim=1) if self.softmax_out else self.act_out(x)
