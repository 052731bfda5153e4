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
  * ZERO-COPY buckets (round 5).  The buckets' slices are registered as GRADIENT DESTINATIONS (``grad_destination``): the
    package's backward kernels -- convolution weight gradients, BatchNorm dgamma / dbeta -- write their result straight into
    the parameter's slice and hand autograd that view, which ``AccumulateGrad`` adopts as ``p.grad`` (no copy: the incoming
    tensor is stolen).  A hook that finds ``p.grad`` aliasing its slice has nothing to pack; the few parameters whose gradient
    arrives as a tensor of its own (biases, parameters with several consumers) are copied as before.  The one-rank RCCL run
    of the benchmark cost 4.7 ms per step for the ~880 per-parameter pack copies in round 4;
  * after a bucket's collective is launched every member's ``p.grad`` is set to None until ``finish()`` (the bucket owns the
    data while the all-reduce runs in place): a second, un-synchronised ``backward()`` then produces fresh tensors that hold
    ONLY the new contribution and cannot disturb the collective; ``finish()`` reduces those separately and adds them (the
    dirty-bucket path).  ``finish()`` re-points ``p.grad`` at the averaged slices: they alias the bucket until the next
    step's backward overwrites it -- a gradient kept across steps must be cloned by the caller (``copy_out=True`` makes
    ``finish()`` hand out independent tensors instead, for gradient-inspection code);
  * the average: ``ReduceOp.AVG`` on RCCL (no scaling pass); SUM + one in-place scale per bucket on backends without it;
  * which parameters are live is agreed on ACROSS ranks (one small MAX all-reduce of a bitmap per step): ranks whose
    losses touched different parameters (a data-dependent branch, a skipped batch) still issue identical collectives
    instead of hanging.  When that agreement says "rebuild" in a step in which some rank still holds old buckets back
    (a gradient that never arrived there), every rank first issues the old buckets it has not started, in index order, so
    that the old layout's collectives pair up on all ranks before the new layout's are issued;
  * a bucket member that NO rank produced a gradient for in this step keeps ``p.grad = None`` (as in the single-GPU
    run: momentum / weight-decay optimizers skip it), everything else receives the average;
  * BatchNorm statistics stay per replica, exactly like N independent runs of the single-GPU reference.
"""
import contextlib
import warnings

import torch
import torch.distributed as dist


def seed_per_rank(base_seed, rank=None):
    """replicas start from IDENTICAL parameters (construct the model under one common seed, then
    ``GradAllReducer`` broadcasts rank 0's) but draw INDEPENDENT dropout masks, auto-mask tie-break noise and augmentation
    parameters: call this after the model is built.  Seeds the torch CPU generator (dropout counter seeds are drawn from it),
    every device generator, numpy and ``random`` with ``base_seed + 1000 * (rank + 1)``; returns that seed."""
    import random

    import numpy as np
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    seed = int(base_seed) + 1000 * (int(rank) + 1)
    torch.manual_seed(seed)
    np.random.seed(seed % (2 ** 31))
    random.seed(seed)
    return seed


# Gradient destinations: parameter storage address -> [view of the parameter's bucket slice, step id of the claim, bucket, index].  The
# package's autograd Functions ask ``grad_destination(param)`` for the tensor to write a parameter's gradient into; the first
# asker of a step gets the slice (if the parameter has no gradient yet), everybody else None (they allocate as usual and
# autograd adds).  Empty unless a GradAllReducer is active: single-process runs are untouched.
_GRAD_DEST = {}
_STEP = [0]
ZERO_COPY = {"taken": 0, "copied": 0}      # diagnostics / tests: bucket members found in place / packed by a copy


def grad_destination(param):
    """-> a dense float32 view shaped like ``param`` to write its gradient into (and to return to autograd), or None"""
    if not _GRAD_DEST:
        return None
    ent = _GRAD_DEST.get(param.data_ptr())
    if ent is None or ent[1] == _STEP[0] or param.grad is not None or ent[0].shape != param.shape:
        return None
    b, i = ent[2], ent[3]
    if b.launched or b.ready[i]:
        # the slice already holds this step's gradient (handed over by a hook, maybe being reduced in place right now) although it
        # was never CLAIMED -- zero_grad(set_to_none=False) leaves p.grad pointing at the slice, so the first backward accumulated
        # into it without asking: a later un-synchronised backward() must not write there (its tensor of its own is
        # picked up by the dirty-bucket path of finish())
        return None
    ent[1] = _STEP[0]
    # a tensor object of its own: AccumulateGrad adopts an incoming gradient without a copy only if nobody else holds it
    return ent[0].view(ent[0].shape)


class _Bucket:
    def __init__(self, params, device):
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += -(-p.numel() // 4) * 4           # every slice starts on 16 bytes: the kernels that write gradients there use 16-byte accesses
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.work = None
        # per-member slices, parameter-shaped views and addresses, made once: the hooks and finish() run per parameter per step
        # (~460 of each for the R101 joint model) and a slice / view_as / data_ptr each time was most of their host cost
        self.slots = [self.flat[o:o + p.numel()] for o, p in zip(self.offsets, params)]
        self.views = [s.view_as(p) for s, p in zip(self.slots, params)]
        self.ptrs = [s.data_ptr() for s in self.slots]
        self.reset()

    def reset(self):
        self.todo = []                              # members whose gradient still has to be copied into its slot (flush_pack)
        self.ready = [False] * len(self.params)
        self.had = [False] * len(self.params)       # packed from a gradient (not a zero fill) this step
        self.pending = len(self.params)
        self.dirty = False
        self.launched = False

    def slot(self, i):
        return self.slots[i]

    def aliased(self, i):
        g = self.params[i].grad
        return g is not None and g.data_ptr() == self.ptrs[i]

    def pack(self, i):
        """member i's gradient into its slot.  A gradient that lives in a tensor of its own (biases, parameters with several
        consumers: ~120 of the R101 joint model's 880) is only NOTED here; ``flush_pack`` moves all of a bucket's in one
        multi-tensor copy right before the collective (one launch per bucket instead of one small copy kernel per parameter)."""
        g = self.params[i].grad
        if g is None:
            self.slots[i].zero_()
            self.todo = [j for j in self.todo if j != i]
        elif g.data_ptr() == self.ptrs[i]:
            self.had[i] = True                      # written in place by its backward kernel (or accumulated in place): nothing to move
            ZERO_COPY["taken"] += 1
        else:
            if i not in self.todo:
                self.todo.append(i)
                ZERO_COPY["copied"] += 1
            self.had[i] = True

    def flush_pack(self):
        if self.todo:
            srcs = [self.params[i].grad for i in self.todo]
            keep = [k for k, g in enumerate(srcs) if g is not None and g.data_ptr() != self.ptrs[self.todo[k]]]
            if keep:
                torch._foreach_copy_([self.views[self.todo[k]] for k in keep], [srcs[k] for k in keep])
            self.todo = []

    def own(self):
        """the collective is about to run in place on ``flat``: the members' p.grad leave the bucket until finish()"""
        self.flush_pack()
        for p in self.params:
            p.grad = None
        self.launched = True


class GradAllReducer:
    def __init__(self, module, bucket_mb=32.0, process_group=None, overlap=True, always=False, control_group=None,
                 timing=False, copy_out=False):
        """control_group: a gloo group over the same ranks as ``process_group`` for the per-step agreement.  None creates
        one with ``dist.new_group`` -- a collective over the DEFAULT group, so every rank of the job must then construct
        its reducer (or pass a pre-created group when only a sub-group trains); False runs the agreement on
        ``process_group`` itself with a device tensor (one small device synchronisation per step) -- the agreement then
        shares a communicator with the bucket all-reduces, so nothing may start from the hooks (ranks that launch a
        different number of buckets before ``finish()`` would pair a bucket with the bitmap): overlap is switched off."""
        # timing=True (bench.py): every bucket's collective is issued from a side stream of its own that does nothing else --
        # it waits for the launch stream (bucket packed), records an event, issues the all-reduce, and records a second event
        # once the collective is done.  ``timing_ms()`` turns the pairs of the last step into the time the collectives
        # occupied (their union: consecutive buckets queue behind each other on the communicator's stream).
        self.timing = bool(timing)
        self.copy_out = bool(copy_out)    # finish() hands out gradients that do not alias the buckets (one copy per parameter)
        self._comm_stream = None
        self._events = []
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        # RCCL averages inside the collective; backends without ReduceOp.AVG (gloo) sum and the bucket is scaled once in place
        self._avg = self.backend == "nccl"
        self._dest_keys = []
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
        self.live_syncs = 0        # bitmap agreements on the live-parameter set
        self._next = 0             # buckets are launched strictly in index order (identical order on every rank)
        # control channel: a gloo group of its own for the per-step agreement (host-side facts only -- which p.grad exist --
        # so no device synchronisation, and its collective cannot interleave differently with the bucket all-reduces on
        # different ranks because it lives on another communicator)
        self._ctl = None
        if self.world > 1 and control_group is not None and control_group is not False:
            self._ctl = control_group
        elif self.world > 1 and control_group is None:
            ranks = dist.get_process_group_ranks(process_group) if process_group is not None else None
            try:
                self._ctl = dist.new_group(ranks=ranks, backend="gloo")
            except Exception as ex:      # no usable TCP interface for gloo: the agreement then runs on the main group
                warnings.warn("GradAllReducer: no gloo control group (%r); the per-step agreement runs on the main group "
                              "(one small device synchronisation per step)" % (ex,))
        if self.world > 1 and self._ctl is None:
            self.overlap = False
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

    def complete_unreachable(self, losses):
        """Between two ``backward()`` calls of a step, right before the LAST one: ``losses`` are the tensors still to be
        back-propagated.  Every bucket member that already holds a gradient (from the calls made under ``no_sync()``) and cannot be
        reached from those losses is complete -- the monodepth decoder and the pose networks once train.py:486 has run -- so it is
        handed over now; buckets that fill up start their all-reduce at once and run under the last backward instead of waiting in
        ``finish()`` behind the hooks that never fire for them (buckets start strictly in index order).  The walk follows
        ``grad_fn.next_functions`` and, through a deferred-trunk gate (functional.TrunkGateFn), the trunk it will back-propagate.
        A member that receives a gradient after all is picked up by the dirty-bucket path: a wrong answer here costs time, not
        correctness."""
        if not self.active or self.buckets is None or not self.overlap or not self._sync:
            return 0
        reach, seen = set(), set()
        stack = [l.grad_fn for l in losses if torch.is_tensor(l) and l.grad_fn is not None]
        while stack:
            fn = stack.pop()
            if fn in seen:
                continue
            seen.add(fn)
            v = getattr(fn, "variable", None)
            if v is not None:
                reach.add(v)
            st = getattr(fn, "st", None)           # TrunkGateFn: the encoder graph behind the gate runs inside that backward
            if st is not None and getattr(st, "roots", None):
                stack.extend(r.grad_fn for r in st.roots if r.grad_fn is not None)
            stack.extend(nf for nf, _ in fn.next_functions if nf is not None)
        n = 0
        for b in self.buckets:
            if b.launched:
                continue
            for i, p in enumerate(b.params):
                if not b.ready[i] and p.grad is not None and p not in reach:
                    b.pack(i)
                    b.ready[i] = True
                    b.pending -= 1
                    n += 1
        self._launch_ready()
        return n

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
        self._next = 0
        self._unregister()
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                if p.dtype == torch.float32 and p.is_contiguous():
                    _GRAD_DEST[p.data_ptr()] = [b.views[i], -1, b, i]
                    self._dest_keys.append((p.data_ptr(), b.views[i]))
        if self.overlap:
            for p in used:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.rebuilds += 1

    def _unregister(self):
        # only the entries that are still THIS reducer's: a replacement reducer over the same parameters may have registered its own
        # slices by the time the garbage collector finalises this one (the reducer sits in a reference cycle through its hooks)
        for k, view in self._dest_keys:
            ent = _GRAD_DEST.get(k)
            if ent is not None and ent[0] is view:
                del _GRAD_DEST[k]
        self._dest_keys = []

    def close(self):
        """stop handing the buckets out as gradient destinations.  MANDATORY before dropping a reducer whose model lives on: the
        destinations are process-global, and a dropped reducer that was never closed keeps receiving the kernels' gradient writes
        in its dead buckets until the garbage collector gets to it"""
        self._unregister()

    def __del__(self):
        try:
            self._unregister()
        except Exception:
            pass

    def _launch(self, b, flat=None):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        buf = b.flat if flat is None else flat
        if flat is None:
            b.own()
        if self.timing and b.flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(b.flat.device)
            cs = self._comm_stream
            cs.wait_stream(torch.cuda.current_stream(b.flat.device))
            b.t0 = torch.cuda.Event(enable_timing=True)
            b.t1 = None
            with torch.cuda.stream(cs):
                b.t0.record(cs)
                b.work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
                if self.backend == "nccl":
                    # RCCL: wait() only makes the side stream wait for the communicator's stream (the host does not block), so
                    # the end event can be queued right here and carries the collective's completion time.  (Queued in
                    # finish() it would carry the time finish() was called: the whole backward, not the collective.)
                    b.work.wait()
                    b.t1 = torch.cuda.Event(enable_timing=True)
                    b.t1.record(cs)
        else:
            b.work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
        self.collectives += 1

    def _wait(self, b):
        """the launch stream (and, for host-side backends, the host) waits for the bucket's collective"""
        if self.timing and b.flat.is_cuda and getattr(b, "t0", None) is not None:
            cs = self._comm_stream
            t1 = b.t1
            if t1 is None:               # host-side backend (gloo): wait() blocks the host, so it happens here
                with torch.cuda.stream(cs):
                    b.work.wait()
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record(cs)
            torch.cuda.current_stream(b.flat.device).wait_stream(cs)
            self._events.append((b.t0, t1, b.flat.numel() * 4))
            b.t0 = b.t1 = None
        else:
            b.work.wait()
        b.work = None

    def timing_ms(self, reset=True):
        """(milliseconds the collectives recorded since the last reset occupied -- the union of their [start, end] intervals
        on the device clock --, number of collectives, bytes); call after a device synchronisation"""
        ev = self._events
        if reset:
            self._events = []
        if not ev:
            return 0.0, 0, 0
        ref = ev[0][0]
        spans = sorted((ref.elapsed_time(a), ref.elapsed_time(b)) for a, b, _ in ev)
        total, cur_s, cur_e = 0.0, spans[0][0], spans[0][1]
        for s_, e_ in spans[1:]:
            if s_ > cur_e:
                total += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        total += cur_e - cur_s
        return total, len(ev), sum(n for _, _, n in ev)

    def _launch_ready(self):
        """Collectives of one group must be issued in the same order on every rank.  Gradient-ready order is not that: a
        rank whose loss skipped a branch completes its buckets in another order (or not at all).  So bucket k starts only
        after buckets 0..k-1 have started; an incomplete one holds the later ones back until finish() fills it in.  In the
        common case (bucket order = reverse registration order = ready order) nothing waits."""
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _on_grad(self, p):
        if not self._sync:
            return
        b, i = self._where[p]
        if b.launched:
            # a second backward() of this step reached a parameter whose bucket is already being reduced: p.grad is a fresh
            # tensor holding this contribution alone (own() took the first one); finish() reduces it separately and adds it
            b.dirty = True
            return
        if b.ready[i]:
            b.pack(i)            # handed over but not launched yet (held back): take the accumulated gradient again
            return
        b.pack(i)
        b.ready[i] = True
        b.pending -= 1
        if b.pending == 0:
            self._launch_ready()

    def finish(self):
        """call after the last backward() of the step, before clipping / the optimizer step"""
        if not self.active:
            return
        cand = [p for p in self.module.parameters() if p.requires_grad]

        def has_grad(p):          # on this rank, this step: still in p.grad, or already taken over by a launched bucket
            if p.grad is not None:
                return True
            w = self._where.get(p)
            return w is not None and w[0].had[w[1]]
        live = [p for p in cand if has_grad(p)]
        rebuild = self.buckets is None or any(p not in self._where for p in live)
        nb = len(self.buckets) if self.buckets is not None else 0
        dirty = [1 if b.dirty else 0 for b in self.buckets] if nb else []
        touched = [1 if has_grad(p) else 0 for p in cand]      # has a gradient THIS step (on this rank)
        if self.world > 1:
            # the decision, the parameter set and the dirty buckets must be the same on every rank, or the collectives
            # diverge and the job hangs: one MAX all-reduce of [rebuild?, ever-had-a-gradient bitmap, has-one-this-step bitmap,
            # dirty bitmap] per step on the control channel (CPU tensor, a few KB).  Unconditional: a rank cannot know that
            # ANOTHER rank saw a new parameter.
            flags = torch.tensor([1 if rebuild else 0] + [1 if (t or p in self._where) else 0 for t, p in zip(touched, cand)]
                                 + touched + dirty, dtype=torch.int32)
            if self._ctl is not None:
                dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self._ctl)
            else:                        # no host-side channel: the main group carries it (device tensor for RCCL)
                dflags = flags.to(cand[0].device) if cand and self.backend == "nccl" else flags
                dist.all_reduce(dflags, op=dist.ReduceOp.MAX, group=self.group)
                flags = dflags.cpu()
            self.live_syncs += 1
            fl = flags.tolist()
            n = len(cand)
            pbits = fl[1:1 + n]
            touched = fl[1 + n:1 + 2 * n]
            rebuild = bool(fl[0]) or any(f and p not in self._where for f, p in zip(pbits, cand))
            union = [p for f, p in zip(pbits, cand) if f]
            dirty = fl[1 + 2 * n:]
        else:
            union = [p for p in cand if has_grad(p) or p in self._where]
        if not union:
            return                       # nothing has a gradient yet on any rank (e.g. a skipped batch on the first step)
        touched_ids = {id(p) for t, p in zip(touched, cand) if t}
        inv = 1.0 / self.world

        def settle(b, d):
            """bucket b's first collective is issued; wait for it, fold in a later backward's contributions (d: dirty on some
            rank), leave the AVERAGE in b.flat"""
            self._wait(b)
            if d:
                if b.dirty and not self._warned:
                    warnings.warn("GradAllReducer: backward() ran more than once in this step outside no_sync(); the later "
                                  "contributions are reduced on their own after the last backward (correct, not overlapped)")
                    self._warned = True
                extra = torch.zeros_like(b.flat)
                for i, p in enumerate(b.params):
                    if p.grad is not None:
                        extra[b.offsets[i]:b.offsets[i] + p.numel()].copy_(p.grad.reshape(-1))
                        b.had[i] = True
                        p.grad = None
                self._launch(b, flat=extra)
                self._wait(b)
                b.flat.add_(extra)
            if not self._avg and inv != 1.0:
                b.flat.mul_(inv)

        if rebuild:
            # first step, or a parameter received its first gradient (on any rank): (re)build the buckets over everything
            # that has ever had a gradient, and reduce this step without overlap
            if self.buckets is not None:
                # the old layout's collectives must pair up on every rank before the new layout's start: a rank that held
                # buckets back (a gradient that did not arrive there) issues them now, in index order, like every other
                # rank did from its hooks.  Their results are KEPT: every rank then holds the old members' average, hands it
                # to the new layout as its "local" gradient (the average of identical values is that value), and only the
                # newcomers are really reduced there.
                for b in self.buckets:
                    if not b.launched:
                        for i in range(len(b.params)):
                            if not b.ready[i]:
                                b.pack(i)
                        self._launch(b)
                for b, d in zip(self.buckets, dirty):
                    settle(b, d)
                    for i, p in enumerate(b.params):
                        p.grad = b.views[i] if id(p) in touched_ids else None
                    b.reset()
                dirty = []
            self._build(union)
        # pass 1, index order like the hooks: whatever has not started yet (incomplete because a gradient did not show up in
        # the last backward, held back behind an incomplete one, overlap off, or fresh after a rebuild)
        for b in self.buckets:
            if not b.launched:
                for i in range(len(b.params)):
                    if not b.ready[i]:
                        b.pack(i)
                self._launch(b)
        # pass 2, after every first-pass launch (same issue order on all ranks): wait; buckets that a second un-synchronised
        # backward reached on ANY rank reduce those later contributions on their own and add them
        dirty = list(dirty) + [0] * (len(self.buckets) - len(dirty))
        for b, d in zip(self.buckets, dirty):
            settle(b, d)
            # every bucket member that had a gradient on SOME rank receives the average on every rank (a parameter that had
            # none locally still gets the other ranks' average: replicas stay identical); one that no rank touched keeps
            # grad = None, as in the single-GPU run.  No copy back: p.grad is re-pointed at its slice of the bucket (valid
            # until the next step's backward writes there); the optimizer reads the bucket.
            for i, p in enumerate(b.params):
                if id(p) in touched_ids:
                    p.grad = b.views[i].clone() if self.copy_out else b.views[i]
                else:
                    p.grad = None
            b.reset()
        self._next = 0
        _STEP[0] += 1                # the destinations may be claimed again
