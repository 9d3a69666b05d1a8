"""Data-parallel gradient synchronisation: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The PICNet path shards over independent samples (no batch-coupled op: encoder norm=none, decoder InstanceNorm), so
the only exchange step is the gradient all-reduce before each optimiser step.  Gradients are
packed into a few large flat buckets (48 MB of G gradients -> 2 buckets): xGMI is point-to-point, large messages
amortise the per-collective latency.  Parameters that received no gradient on any rank (Auto_Attn.alpha / .model.*,
base_function.py:410-418) are skipped consistently because the set is a function of the graph, not of the data.
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


_DEBUG = bool(os.environ.get("FMI_DIST_DEBUG"))

# How a gradient bucket is summed over the ranks (FMI_DP_EXCHANGE, or the ``exchange`` argument of DataParallelOptimizer):
#   "allreduce"  one all-reduce per bucket (the library picks the algorithm; a ring is bound by ONE xGMI link: 2 (n-1)/n S per link)
#   "rs_ag"      reduce-scatter + all-gather as two collectives (S / n per link and phase when the library runs them direct)
#   "direct"     the same exchange spelled out point-to-point: all-to-all of the n shards (every rank sends shard j straight to rank j over
#                its own link: all 7 links of a fully connected node carry S / n at once), a local sum, all-gather of the reduced
#                shards -- SURVEY.md 5.8 / 8e.  Also runs on gloo, so the CPU tests cover it.
# Unmeasured on hardware: no multi-GPU node has been available to any round so far; the flag exists so that all three can be timed.
EXCHANGES = ("allreduce", "rs_ag", "direct")


def default_exchange() -> str:
    e = os.environ.get("FMI_DP_EXCHANGE", "allreduce")
    if e not in EXCHANGES:
        raise ValueError(f"FMI_DP_EXCHANGE={e!r}: one of {EXCHANGES}")
    return e


class _Pending:
    """one bucket on the wire: wait() leaves the SUM over ranks in ``flat[:n]``"""

    def __init__(self, flat, n, works, shards=None, out=None):
        self.flat, self.n, self.works, self.shards, self.out = flat, n, works, shards, out

    def wait(self):
        for w in self.works:
            w.wait()
        return self.flat[:self.n]


def exchange_sum(flat: torch.Tensor, how: str, group=None) -> _Pending:
    """start summing ``flat`` over the ranks, asynchronously"""
    world = dist.get_world_size(group)
    n = flat.numel()
    if how == "allreduce" or world == 1:
        return _Pending(flat, n, [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)])
    per = (n + world - 1) // world
    if per * world != n:  # shards of equal length
        flat = torch.cat([flat, flat.new_zeros(per * world - n)])
    if how == "rs_ag" and dist.get_backend(group) == "nccl":
        shard = flat.new_empty(per)
        w1 = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        w2 = dist.all_gather_into_tensor(flat, shard, group=group, async_op=True)  # stream-ordered behind the reduce-scatter
        return _Pending(flat, n, [w1, w2], shards=shard)
    # "direct" (and "rs_ag" on a backend without reduce-scatter): shard j of every rank goes straight to rank j
    recv = flat.new_empty(per * world)
    dist.all_to_all_single(recv, flat, group=group)
    mine = recv.view(world, per).sum(0)
    return _Pending(flat, n, [dist.all_gather_into_tensor(flat, mine, group=group, async_op=True)], shards=mine, out=recv)


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_parameters(modules: Iterable[torch.nn.Module], src: int = 0) -> None:
    """make every replica start from rank 0's parameters AND SpectralNorm u/v state"""
    if not is_distributed():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)


def allreduce_gradients(params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20, group=None) -> int:
    """average .grad over ranks in flat buckets; returns the number of collectives issued"""
    if not is_distributed():
        return 0
    world = dist.get_world_size(group)
    grads: List[torch.Tensor] = [p.grad for p in params if p.grad is not None]
    ncoll, bucket, size = 0, [], 0

    def flush():
        nonlocal ncoll, bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        ncoll += 1
        bucket, size = [], 0

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    return ncoll


class DataParallelOptimizer:
    """Wraps an optimiser so that ``step()`` applies the gradient averaged over all ranks.  Drop-in for the
    ``optimizer_G`` / ``optimizer_D`` arguments of GANOptimizer (loss.py:70).

    The all-reduce is overlapped with the backward pass: a post-accumulate hook on every parameter appends its finished
    gradient to the open bucket, and a full bucket is flattened and all-reduced asynchronously (RCCL runs it on the process
    group's own stream) while autograd keeps producing the earlier layers' gradients.  ``launch()`` sends whatever is still
    open without blocking -- GANOptimizer uses it to put the discriminator's gradients on the wire before the generator
    backward (whose first ~20 % is the VGG dgrad, which produces no gradients of its own) starts.  ``step()`` sends the
    rest, waits, writes the averages back into ``.grad`` and steps.  Bucket membership follows hook order, which is a
    function of the graph and therefore identical on every rank.  One backward per step (no gradient accumulation)."""

    def __init__(self, optimizer: torch.optim.Optimizer, bucket_bytes: int = 16 << 20, group=None, exchange=None):
        self.optimizer = optimizer
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.exchange = exchange or default_exchange()
        if self.exchange not in EXCHANGES:
            raise ValueError(f"exchange={self.exchange!r}: one of {EXCHANGES}")
        self._open: List[torch.nn.Parameter] = []
        self._open_bytes = 0
        self._inflight = []  # (work, flat, params)
        self.collectives = 0
        self._hooks = []
        if is_distributed():
            for g in optimizer.param_groups:
                for p in g["params"]:
                    if p.requires_grad:
                        self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        self._open.append(p)
        self._open_bytes += p.grad.numel() * p.grad.element_size()
        if self._open_bytes >= self.bucket_bytes:
            self.launch()

    def launch(self):
        """all-reduce the open bucket asynchronously; returns immediately"""
        if not self._open:
            return
        params, self._open, self._open_bytes = self._open, [], 0
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        if _DEBUG:
            import sys
            print(f"[fmi.dist r{dist.get_rank()}] bucket #{self.collectives} of {type(self.optimizer).__name__}@{id(self) & 0xffff:x}: "
                  f"{len(params)} tensors, {flat.numel() * 4} bytes", file=sys.stderr, flush=True)
        work = exchange_sum(flat, self.exchange, self.group)
        self._inflight.append((work, flat, params, [p.grad.numel() for p in params]))
        self.collectives += 1

    def _finish(self):
        self.launch()
        world = dist.get_world_size(self.group)
        for work, flat, params, sizes in self._inflight:
            flat = work.wait()
            flat.div_(world)
            dst, views, off = [], [], 0
            for p, n in zip(params, sizes):
                if p.grad is not None and p.grad.numel() == n:  # a gradient cleared (set_to_none) after its bucket left is skipped
                    dst.append(p.grad)
                    views.append(flat[off:off + n].view_as(p.grad))
                off += n
            if dst:
                torch._foreach_copy_(dst, views)
        self._inflight = []

    def _drop_inflight(self):
        """a backward fired the hooks but no step() followed (skipped step, exception, validation backward): the collectives
        must still complete on every rank, their results are discarded"""
        for work, _flat, _params, _sizes in self._inflight:
            work.wait()
        self._inflight = []

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def zero_grad(self, *a, **k):
        self._open, self._open_bytes = [], 0
        self._drop_inflight()
        return self.optimizer.zero_grad(*a, **k)

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, sd):
        return self.optimizer.load_state_dict(sd)

    def step(self, closure=None):
        if is_distributed():
            if self._hooks:
                self._finish()
            else:  # process group created after this wrapper: plain bucketed all-reduce
                allreduce_gradients([p for g in self.optimizer.param_groups for p in g["params"]], self.bucket_bytes, self.group)
        return self.optimizer.step(closure)
