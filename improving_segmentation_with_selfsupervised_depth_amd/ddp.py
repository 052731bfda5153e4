"""Data-parallel gradient averaging for one process per GPU (RCCL over xGMI through torch.distributed 'nccl';
'gloo' in the CPU tests).  The reference has no distributed path at all: this is the one real
exchange step of the training path -- a bucketed all-reduce of the gradients, launched from post-accumulate hooks
so that it overlaps with the rest of the backward pass.

Design notes
  * buckets are built in reverse registration order (the order gradients become ready), ~32 MB fp32 each: 396 MB
    of gradients for the R101 joint model is ~13 collectives -- large enough that the per-link xGMI ring is
    bandwidth- not latency-bound, small enough to start while the encoder backward still runs;
  * parameters that never receive a gradient (requires_grad=False, and the dead dispconv heads inside the
    segmentation decoder's private DepthDecoder) are detected on the first step and left out of the buckets; a
    parameter that receives its first gradient on a later step (another loss branch switched on, a sub-model
    unfrozen) makes ``finish()`` rebuild the buckets with it included -- nothing is ever left un-reduced;
  * the reference's train step calls ``backward()`` up to five times (train.py:486-510, 698, 724) and lets the
    gradients accumulate.  Collectives may only start during the LAST of them: wrap the earlier ones in
    ``with reducer.no_sync():`` (hooks then do nothing).  A step that forgets to do so is still reduced correctly --
    a hook that fires twice for one parameter marks its bucket dirty and ``finish()`` re-reduces it from the
    accumulated ``p.grad`` -- it only loses the overlap (and warns once);
  * the averaged bucket is scaled once in place and copied back with one multi-tensor launch per bucket (the flat
    buffers are never aliased by ``p.grad``: an in-flight collective cannot be disturbed by a later accumulation);
  * BatchNorm statistics stay per replica, exactly like N independent runs of the single-GPU reference.
"""
import contextlib
import warnings

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, device):
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.work = None
        self.reset()

    def reset(self):
        self.ready = [False] * len(self.params)
        self.pending = len(self.params)
        self.dirty = False

    def slot(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()]

    def pack(self, i):
        p = self.params[i]
        s = self.slot(i)
        if p.grad is None:
            s.zero_()
        else:
            s.copy_(p.grad.reshape(-1))


class GradAllReducer:
    def __init__(self, module, bucket_mb=32.0, process_group=None, overlap=True, always=False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self.always = always       # run the bucket / hook / collective machinery even with a single rank (self-test)
        self.bucket_bytes = int(bucket_mb * 1024 * 1024)
        self.overlap = overlap
        self.buckets = None
        self._where = {}
        self._hooks = []
        self._sync = True
        self._warned = False
        self.collectives = 0       # diagnostics / tests: all_reduce launches so far
        self.rebuilds = 0
        self.broadcast_parameters()

    @property
    def active(self):
        return self.world > 1 or self.always

    def broadcast_parameters(self):
        if not self.active:
            return
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t, 0, group=self.group)

    @contextlib.contextmanager
    def no_sync(self):
        """for every backward() of a step except the last one: gradients accumulate locally, no collective starts"""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _build(self, used):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        device = used[0].device
        self.buckets, cur, size = [], [], 0
        for p in reversed(used):
            cur.append(p)
            size += p.numel() * 4
            if size >= self.bucket_bytes:
                self.buckets.append(_Bucket(cur, device))
                cur, size = [], 0
        if cur:
            self.buckets.append(_Bucket(cur, device))
        self._where = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
        if self.overlap:
            for p in used:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.rebuilds += 1

    def _launch(self, b):
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.collectives += 1

    def _on_grad(self, p):
        if not self._sync:
            return
        b, i = self._where[p]
        if b.ready[i] or b.work is not None:
            # a second backward() of this step reached a parameter whose gradient was already handed over: what sits
            # in the bucket is stale.  finish() re-reduces this bucket from the accumulated p.grad.
            b.dirty = True
            return
        b.pack(i)
        b.ready[i] = True
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def finish(self):
        """call after the last backward() of the step, before clipping / the optimizer step"""
        if not self.active:
            return
        live = [p for p in self.module.parameters() if p.requires_grad and p.grad is not None]
        if self.buckets is None or any(p not in self._where for p in live):
            # first step, or a parameter received its first gradient: (re)build the buckets over everything that has
            # ever had a gradient, and reduce this step without overlap
            if self.buckets is not None:
                for b in self.buckets:
                    if b.work is not None:
                        b.work.wait()
                        b.work = None
            known = set(self._where)
            used = [p for p in self.module.parameters() if p.requires_grad and (p.grad is not None or p in known)]
            self._build(used)
        for b in self.buckets:
            if b.dirty:
                if not self._warned:
                    warnings.warn("GradAllReducer: backward() ran more than once in this step outside no_sync(); the "
                                  "affected buckets are reduced again after the last backward (correct, not overlapped)")
                    self._warned = True
                if b.work is not None:
                    b.work.wait()
                    b.work = None
                b.reset()
            if b.work is None:   # incomplete (a gradient did not show up in the last backward), dirty, or overlap off
                for i in range(len(b.params)):
                    if not b.ready[i]:
                        b.pack(i)
                self._launch(b)
        inv = 1.0 / self.world
        for b in self.buckets:
            b.work.wait()
            b.work = None
            if inv != 1.0:
                b.flat.mul_(inv)
            # identical programs on every rank (pure data parallelism): a parameter without a gradient
            # this step has none on any rank and keeps grad=None, as in the single-GPU reference
            idx = [i for i, p in enumerate(b.params) if p.grad is not None]
            if idx:
                torch._foreach_copy_([b.params[i].grad for i in idx], [b.slot(i).view_as(b.params[i]) for i in idx])
            b.reset()
