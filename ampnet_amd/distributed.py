"""One-subgraph-per-GPU data parallelism: gradient averaging over RCCL.

Mirrors what experiments/cora_benchmark_graphsaint_distributed.py:63-66,83-94
INTENDS (DDP over per-rank GraphSAINT subgraphs; the script calls the un-wrapped
model at :83 so its reducer never arms -- SURVEY.md section 3.3).  Here every rank
draws its own subgraph, runs forward/backward locally, and one all-reduce of ONE
flat fp32 buffer (<= ~2 MB for two AMPConv layers at D=256: latency-bound, so a
single collective, no bucketing) averages the gradients before optimizer.step().

Backend "nccl" is RCCL on ROCm (xGMI between the 8 MI355X of a node); "gloo" is
used by the CPU tests.
"""
import torch
import torch.distributed as dist


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (what
    DistributedDataParallel does at construction, ..._distributed.py:63)."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class GradientAllReducer:
    """Flat-buffer mean all-reduce of `params`' gradients.

        reducer = GradientAllReducer(model.parameters())
        loss.backward(); reducer.allreduce(); optimizer.step()
    """

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self._flat = None

    def _buffer(self, like):
        if self._flat is None or self._flat.device != like.device:
            self._flat = torch.empty(self.numel, dtype=torch.float32, device=like.device)   # fp32 sum
            self._views, off = [], 0
            for p in self.params:
                self._views.append(self._flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        return self._flat

    def allreduce(self, average=True):
        """average=True: mean over ranks (independent mini-batches per rank); False: sum (ranks hold
        partitions of ONE graph, ampnet_amd/partitioned.py).  Pack and unpack are ONE multi-tensor copy each
        (the payload is ~1 MB: on a 34 ms GraphSAINT step 2 x #parameters tiny launches would sit on the
        critical path) and the 1/world factor rides on the unpack."""
        if not self.params:
            return None
        world = dist.get_world_size(self.group)
        flat = self._buffer(self.params[0])
        have = [i for i, p in enumerate(self.params) if p.grad is not None]
        if len(have) != len(self.params):
            flat.zero_()                                # missing grads count as zero
        if have:
            torch._foreach_copy_([self._views[i] for i in have], [self.params[i].grad for i in have])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)      # also as the only rank: the same code path
        if average and world > 1:
            flat.mul_(1.0 / world)
        for i, p in enumerate(self.params):
            if p.grad is None:
                p.grad = torch.empty_like(p)
        torch._foreach_copy_([p.grad for p in self.params], self._views)
        return flat
