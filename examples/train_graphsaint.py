#!/usr/bin/env python3
"""The reference's main training harness (experiments/cora_benchmark_graphsaint.py:59-135) on the
MI355X path, on a synthetic Cora-shaped graph (Cora itself is a network download):
AMPGCN(D=128, H=4, L=20) + GraphSAINT random-walk batches + Adam + cosine warm restarts +
node_norm-weighted NLL.  Everything between the data and the loss runs on the GPU.

    python examples/train_graphsaint.py [--epochs 3]
"""
import argparse
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from ampnet_amd import AMPGCN, GraphSAINTRandomWalkSampler  # noqa: E402


def synthetic_cora(device, n=2708, f=1433, classes=7, seed=1):
    """Bag-of-words-like features whose present words depend on the class; homophilous edges."""
    g = torch.Generator().manual_seed(seed)
    y = torch.randint(0, classes, (n,), generator=g)
    topic = torch.rand(classes, f, generator=g) < 0.03                  # class vocabulary
    x = ((torch.rand(n, f, generator=g) < 0.004) | (topic[y] & (torch.rand(n, f, generator=g) < 0.3))).float()
    x[torch.arange(n), torch.randint(0, f, (n,), generator=g)] = 1.0    # at least one present word
    src = torch.randint(0, n, (5278,), generator=g)
    same = torch.rand(5278, generator=g) < 0.8                           # 80 % intra-class edges
    perm = torch.argsort(y + torch.rand(n, generator=g) * 0.5)
    pos = torch.empty(n, dtype=torch.long); pos[perm] = torch.arange(n)
    near = perm[(pos[src] + torch.randint(1, 40, (5278,), generator=g)).clamp(max=n - 1)]
    dst = torch.where(same, near, torch.randint(0, n, (5278,), generator=g))
    ei = torch.cat([torch.stack([src, dst]), torch.stack([dst, src])], dim=1)   # both directions
    idx = torch.randperm(n, generator=g)
    mask = lambda a, b: torch.zeros(n, dtype=torch.bool).index_fill_(0, idx[a:b], True)
    return types.SimpleNamespace(x=x.to(device), y=y.to(device), edge_index=ei.to(device), num_nodes=n,
                                 train_mask=mask(0, 1400).to(device), test_mask=mask(1400, n).to(device))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--class-defaults', action='store_true',
                    help="AMPGCN's class defaults (src/ampnet/module/amp_gcn.py:21-35: embedding_dim=100, heads=2, 40 sampled "
                         "vectors: what experiments/cora_benchmark_graphsaint_distributed.py:58 instantiates) instead of the "
                         "128 / 4 / 20 of experiments/cora_benchmark_graphsaint.py")
    args = ap.parse_args()
    device = torch.device('cuda:0')
    torch.manual_seed(1)
    data = synthetic_cora(device)
    D, H, L = (100, 2, 40) if args.class_defaults else (128, 4, 20)
    model = AMPGCN(device=device, embedding_dim=D, num_heads=H, num_node_features=1433, num_sampled_vectors=L,
                   output_dim=7, softmax_out=True, feat_emb_dim=D - 1, val_emb_dim=1, dropout_rate=0.0,
                   dropout_adj_rate=0.0).to(device)
    loader = GraphSAINTRandomWalkSampler(data, batch_size=8, walk_length=150, num_steps=args.steps,
                                         sample_coverage=20, seed=1)
    opt = torch.optim.Adam(model.parameters(), lr=0.005, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=400, T_mult=2)
    t0 = time.time()
    history = []
    for epoch in range(args.epochs):
        tot = cnt = correct = 0
        torch.cuda.synchronize()
        te = time.time()
        for batch in loader:
            model.train()
            opt.zero_grad()
            out = model(batch)
            loss = (F.nll_loss(out, batch.y, reduction='none') * batch.node_norm)[batch.train_mask].sum()
            loss.backward()
            opt.step()
            sched.step()
            tot += loss.item(); cnt += 1
            correct += float((out.argmax(1) == batch.y)[batch.train_mask].float().mean())
        history.append((tot / cnt, correct / cnt))
        torch.cuda.synchronize()
        print(f'epoch {epoch}: train loss {tot / cnt:.4f}  train acc {correct / cnt:.3f}  '
              f'({time.time() - t0:.1f} s; this epoch {time.time() - te:.3f} s = {1e3 * (time.time() - te) / cnt:.2f} ms '
              f'per sampled batch, sampler + 2 AMPConv layers fwd+bwd + Adam)', flush=True)
    model.eval()
    with torch.no_grad():
        out = model(data)                                                   # full-graph eval (:159-163)
        acc = float((out.argmax(1) == data.y)[data.test_mask].float().mean())
    print(f'full-graph test accuracy {acc:.3f}')
    return history, acc


if __name__ == '__main__':
    main()
