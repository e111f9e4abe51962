"""Data-parallel training support: one process per GPU, gradients averaged with one bucketed all-reduce per
network branch over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" on CPU for the tests).

The reference has no distributed code; this is the capability BASELINE configs 3-4 add.  The batch
shards over ranks; the loss is a mean over the local batch, so averaging local gradients over equal shards
reproduces the global-batch gradient exactly.  Parameters that receive no gradient under the
current flags (damping, conv_dec3.*, conf*; SURVEY B-8) are skipped consistently on every rank.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Equal contiguous shard of the leading (batch) dimension."""
    B = t.shape[0]
    if B % world:
        raise ValueError(f'batch {B} does not divide over {world} ranks')
    n = B // world
    return t[rank * n:(rank + 1) * n]


class GradSync:
    """Bucketed gradient averaging, one bucket per network branch.

    ``start(grads, flat)``: ``flat`` is the contiguous fp32 buffer the branch's wgrad kernels wrote their results into
    (``vgg_backward_nhwc(..., flat=True)``; every tensor of ``grads`` that lies inside it is a view of it): it is summed IN
    PLACE by one asynchronous all-reduce -- no gather copy, no scatter copy.  Tensors of ``grads`` outside ``flat`` (the
    confidence heads, level 4's padded layers, ``damping``) and callers without a flat buffer go through a small staged
    bucket.  ``finish(handle)`` waits (RCCL: the collective itself averages, ``ReduceOp.AVG``; gloo: sum, then one division).  Used by the model's backward (``model.grad_sync``) so
    that the satellite branch's 9.9 MB bucket is in flight on the xGMI links while the ground branch's backward kernels run."""

    def __init__(self, group=None, force=False):
        """``force``: issue the collectives even in a one-rank group (how a 1-GPU box exercises the RCCL path)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force) and dist.is_initialized()
        # RCCL averages inside the collective (ncclAvg): no div_ pass over the 19.8 MB afterwards.  gloo has no AVG.
        self.avg_in_collective = dist.is_initialized() and dist.get_backend(group) == 'nccl'
        self.op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
        self.bytes_reduced = 0
        self.collectives = 0
        # evidence hook (bench.py, N > 1): with ``timing`` on, every bucket records an event on the launch stream where its
        # all-reduce is issued, and ``finish`` one before its first wait (= the backward's kernels are all enqueued behind it) and one
        # behind its last; ``overlap_report()`` turns them into "how long before the backward's end each bucket started" and
        # "how long the launch stream then still waited for the collectives" (allreduce_exposed_ms).  CUDA tensors only.
        self.timing = False
        self._timeline = []      # per start(): [(event, bytes)], per finish(): (pre, post)

    @staticmethod
    def _inside(t: torch.Tensor, flat: torch.Tensor) -> bool:
        return (t.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() and t.is_contiguous()
                and flat.data_ptr() <= t.data_ptr() and t.data_ptr() + t.numel() * 4 <= flat.data_ptr() + flat.numel() * 4)

    def start(self, grads: dict, flat: torch.Tensor | None = None):
        if (self.world == 1 and not self.force) or not grads:
            return None
        works = []
        rest = grads
        if flat is not None:
            rest = {n: g for n, g in grads.items() if not self._inside(g, flat)}
            self.bytes_reduced += flat.numel() * 4
            self.collectives += 1
            self._mark_start(flat)
            works.append((dist.all_reduce(flat, op=self.op, group=self.group, async_op=True), flat, None, None))
        if rest:
            names = sorted(rest)                      # identical order on every rank
            stage = torch.cat([rest[n].reshape(-1).float() for n in names])
            self.bytes_reduced += stage.numel() * 4
            self.collectives += 1
            self._mark_start(stage)
            works.append((dist.all_reduce(stage, op=self.op, group=self.group, async_op=True), stage, names, rest))
        return works

    def _mark_start(self, buf: torch.Tensor):
        if self.timing and buf.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._timeline.append(('start', ev, buf.numel() * 4))

    def _mark(self, kind: str, buf: torch.Tensor):
        if self.timing and buf.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._timeline.append((kind, ev, 0))

    def overlap_report(self):
        """After a device synchronisation: per training step (= one run of start()s followed by finish()es) the buckets' issue
        times relative to the first finish (negative: the collective was in flight under that much of the backward) and the
        time the launch stream spent between the first wait and the last one returning.  Clears the record."""
        steps, cur = [], {'starts': [], 'pre': None, 'post': None}
        for kind, ev, nbytes in self._timeline:
            if kind == 'start':
                if nbytes < 4096:      # (a trainable `damping`: four floats reduced behind the two buckets -- part of the same step)
                    continue
                if cur['post'] is not None:
                    steps.append(cur)
                    cur = {'starts': [], 'pre': None, 'post': None}
                cur['starts'].append((ev, nbytes))
            elif kind == 'pre' and cur['pre'] is None:
                cur['pre'] = ev
            elif kind == 'post':
                cur['post'] = ev
        if cur['post'] is not None:
            steps.append(cur)
        self._timeline = []
        out = []
        for st in steps:
            if st['pre'] is None:
                continue
            out.append({'exposed_ms': st['pre'].elapsed_time(st['post']),
                        'bucket_issue_ms_before_backward_end': [round(ev.elapsed_time(st['pre']), 3) for ev, _ in st['starts']],
                        'bucket_bytes': [b for _, b in st['starts']]})
        return out

    def finish(self, handle):
        if handle is None:
            return
        if handle:
            self._mark('pre', handle[0][1])
        for work, buf, names, grads in handle:
            work.wait()
            if not self.avg_in_collective:
                buf.div_(self.world)
            if names is None:
                continue                              # reduced in place: the gradient tensors ARE views of buf
            o = 0
            for n in names:
                g = grads[n]
                g.copy_(buf[o:o + g.numel()].view_as(g))
                o += g.numel()
        if handle:
            self._mark('post', handle[-1][1])


def allreduce_module_grads(module: torch.nn.Module, group=None):
    """Fallback for code that did not install ``grad_sync``: average every existing ``.grad`` after backward."""
    gs = GradSync(group)
    grads = {n: p.grad for n, p in module.named_parameters() if p.grad is not None}
    gs.finish(gs.start(grads))
    return gs.bytes_reduced


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
