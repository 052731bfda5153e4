"""Data-parallel gradient averaging for one process per GPU (RCCL over xGMI through torch.distributed 'nccl';
'gloo' in the CPU tests).  The reference has no distributed path at all: this is the one real
exchange step of the training path -- a bucketed all-reduce of the gradients, launched from post-accumulate hooks
so that it overlaps with the rest of the backward pass.

Design notes
  * buckets are built in reverse registration order (the order gradients become ready), ~32 MB fp32 each: 396 MB
    of gradients for the R101 joint model is ~13 collectives -- large enough that the per-link xGMI ring is
    bandwidth- not latency-bound, small enough to start while the encoder backward still runs;
  * parameters that never receive a gradient (requires_grad=False, and the dead dispconv heads inside the
    segmentation decoder's private DepthDecoder) are detected on the first step and left out of the buckets;
  * BatchNorm statistics stay per replica, exactly like N independent runs of the single-GPU reference.
"""
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
        self.pending = len(params)
        self.work = None


class GradAllReducer:
    def __init__(self, module, bucket_mb=32.0, process_group=None, overlap=True, always=False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.always = always       # run the bucket / hook / collective machinery even with a single rank (self-test)
        self.bucket_bytes = int(bucket_mb * 1024 * 1024)
        self.overlap = overlap
        self.buckets = None
        self._where = {}
        self._hooks = []
        self.broadcast_parameters()

    def broadcast_parameters(self):
        if self.world == 1 and not self.always:
            return
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t, 0, group=self.group)

    def _build(self, used):
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

    def _on_grad(self, p):
        b, i = self._where[p]
        o = b.offsets[i]
        b.flat[o:o + p.numel()].copy_(p.grad.reshape(-1))
        b.pending -= 1
        if b.pending == 0:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """call after backward(), before clipping / the optimizer step"""
        if self.world == 1 and not self.always:
            return
        if self.buckets is None:
            # first step: learn which parameters actually receive gradients, reduce them unbucketed-overlap
            used = [p for p in self.module.parameters() if p.requires_grad and p.grad is not None]
            self._build(used)
            for b in self.buckets:
                for i, p in enumerate(b.params):
                    o = b.offsets[i]
                    b.flat[o:o + p.numel()].copy_(p.grad.reshape(-1))
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        for b in self.buckets:
            if b.work is None:   # a gradient did not show up this step (or overlap is off): reduce what is there
                for i, p in enumerate(b.params):
                    o = b.offsets[i]
                    if p.grad is None:
                        b.flat[o:o + p.numel()].zero_()
                    elif b.pending > 0 or not self.overlap:
                        b.flat[o:o + p.numel()].copy_(p.grad.reshape(-1))
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        inv = 1.0 / self.world
        for b in self.buckets:
            b.work.wait()
            for i, p in enumerate(b.params):
                o = b.offsets[i]
                g = b.flat[o:o + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.mul(inv)
                else:
                    p.grad.copy_(g).mul_(inv)
            b.pending = len(b.params)
            b.work = None
