"""numpy restatement of the deterministic parts of the reference's (vendored PyG) GraphSAINT
sampler -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/visualization/visualize_graphsaint_subgraphs.py:
  :107-110  node_idx = unique(sampled nodes); adj.saint_subgraph(node_idx)  -> induced_subgraph()
  :112-135  relabelled edge_index + original edge ids                        -> induced_subgraph()
  :165-171  edge_norm / node_norm from the occurrence counts                 -> norms()
Parity status of the random walk itself: UNPINNED -- it is torch_sparse's `random_walk` C++ op
(third party, not installed, no fixture in the reference); walk_is_valid() checks the defining
property instead (every step moves along an out-edge, or stays at a node without out-edges).
"""
import numpy as np


def induced_subgraph(edge_index, num_nodes, sampled_nodes):
    node_idx = np.unique(np.asarray(sampled_nodes).reshape(-1))
    relabel = -np.ones(num_nodes, dtype=np.int64)
    relabel[node_idx] = np.arange(node_idx.size)
    src, dst = edge_index
    keep = np.nonzero((relabel[src] >= 0) & (relabel[dst] >= 0))[0]
    return node_idx, np.stack([relabel[src[keep]], relabel[dst[keep]]]), keep


def norms(node_count, edge_count, edge_src, num_nodes, num_samples):
    node_count = np.asarray(node_count, dtype=np.float32).copy()
    edge_count = np.asarray(edge_count, dtype=np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        edge_norm = np.clip(node_count[edge_src] / edge_count, 0, 1e4).astype(np.float32)
    edge_norm[np.isnan(edge_norm)] = 0.1
    node_count[node_count == 0] = 0.1
    node_norm = (np.float32(num_samples) / node_count / np.float32(num_nodes)).astype(np.float32)
    return node_norm, edge_norm


def walk_is_valid(edge_index, num_nodes, walks):
    src, dst = edge_index
    has_out = np.zeros(num_nodes, dtype=bool)
    has_out[src] = True
    edges = set(zip(src.tolist(), dst.tolist()))
    for w in np.asarray(walks):
        for a, b in zip(w[:-1], w[1:]):
            if has_out[a]:
                if (int(a), int(b)) not in edges:
                    return False
            elif a != b:
                return False
    return True
