"""Reference-shaped torch CPU restatement of AMPConv (edge-materialising).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Same op sequence as the
reference: gather x[dst], x[src] -> reshape [E, L, D] -> stock
torch.nn.MultiheadAttention(query=dst tokens, key=value=src tokens) ->
scatter-mean to N rows.  It is what bench.py times as `cpu_baseline`
(kind "port") and what supplies autograd gradients for spot checks at sizes
the golden fixtures do not cover.

Follows (paths relative to /root/reference):
  * src/ampnet/conv/amp_conv.py:10-22   ctor: nn.MultiheadAttention(D, H, batch_first=True, bias=True)
  * src/ampnet/conv/amp_conv.py:24-26   forward -> propagate
  * src/ampnet/conv/amp_conv.py:28-51   message
  * PyG MessagePassing(aggr='mean') semantics, pinned by
    synthetic_benchmark/testing_message_passing_pyg.py:37-40
"""
import torch
import torch.nn as nn


def scatter_mean(msg, index, dim_size):
    out = torch.zeros(dim_size, msg.size(1), dtype=msg.dtype, device=msg.device)
    out.index_add_(0, index, msg)
    cnt = torch.zeros(dim_size, dtype=msg.dtype, device=msg.device)
    cnt.index_add_(0, index, torch.ones_like(index, dtype=msg.dtype))
    return out / cnt.clamp(min=1).unsqueeze(-1)


class RefShapedAMPConv(nn.Module):
    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.attn_output_weights = None
        self.attn_output = None
        self.num_heads = num_heads
        self.embed_dim = embed_dim
        self.multi_head_attention = nn.MultiheadAttention(
            embed_dim=embed_dim, num_heads=num_heads, batch_first=True, bias=True)

    def forward(self, x, edge_index):
        src, dst = edge_index[0], edge_index[1]
        x_i = x.index_select(0, dst)
        x_j = x.index_select(0, src)
        msg = self.message(x_i, x_j)
        return scatter_mean(msg, dst, x.size(0))

    def message(self, x_i, x_j):
        D = self.embed_dim
        if x_i.shape[1] % D != 0:
            raise ValueError("invalid configuration")
        L = x_i.shape[1] // D
        q = x_i.reshape(x_i.shape[0], L, D)
        kv = x_j.reshape(x_j.shape[0], L, D)
        self.attn_output, self.attn_output_weights = self.multi_head_attention(
            query=q, key=kv, value=kv)
        return self.attn_output.reshape(x_i.shape[0], x_i.shape[1])
