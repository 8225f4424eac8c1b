"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI overlapped with backward.

Replaces the reference's single-process `nn.DataParallel` wrapper (train.py:153-213,266), which re-broadcasts all
816 tensors every step and reduces gradients onto GPU 0.  Here every rank owns a full replica; the only collective
is a mean all-reduce of the gradients, issued bucket by bucket on a dedicated HIP stream while the rest of the
backward is still running on the compute stream:

  * buckets are laid out in REVERSE registration order (depth decoder / heads first, backbone stem last), which
    is the order autograd produces the gradients in;
  * a per-parameter post-accumulate-grad hook counts arrivals; when a bucket is complete the compute stream records
    an event, the side stream waits on it, packs the gradients into the bucket's flat buffer, all-reduces it
    (RCCL ring/tree over the xGMI links; ~25 MB buckets keep each collective bandwidth-bound rather than
    latency-bound, and 229 MB total is ~10 launches) and unpacks the mean back into the .grad tensors;
  * `finish()` makes the compute stream wait for the side stream before the optimizer reads the gradients.

BatchNorm statistics stay per-rank, as in the reference's per-replica DataParallel; the loss used for logging is
the mean over ranks (train.py:348).  With world_size == 1 nothing is registered and nothing is launched.
"""
import torch
import torch.distributed as dist


import os as _os
_PROF = {} if _os.environ.get("PRN_EXCHANGE_PROF") else None      # host time inside the hooks / launches / finish (debugging aid)


ALIAS_GRADS = bool(int(_os.environ.get("PRN_EXCHANGE_ALIAS_GRADS", "1")))      # 0: copy the reduced buckets back into the gradient tensors


class GradAllReduce:
    def __init__(self, params, bucket_bytes=25 << 20, process_group=None, force=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets, self._handles, self._pending = [], [], []
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())   # force: exercise the path with one rank (tests)
        if not self.active:
            return
        dev = self.params[0].device
        self.on_gpu = dev.type == "cuda"
        # RCCL averages inside the collective (ncclAvg): no separate division pass over the 229 MB of gradients.  gloo (the CPU tests) has no AVG.
        self.avg_in_collective = self.on_gpu and dist.get_backend(process_group) == "nccl" and not _os.environ.get("PRN_EXCHANGE_NO_AVG")
        self.main = torch.cuda.current_stream(dev) if self.on_gpu else None      # the stream forward / backward are issued on
        # The exchange shares the weight-gradient side stream (ops.set_wgrad_async): most of a bucket is produced there, the
        # side stream is never on the critical path, and every extra HIP stream is one more customer for the four hardware
        # queues (a dedicated exchange stream cost 7 ms/step with ONE rank: 59.7 -> 66.9 ms, see ops.BRANCH_STREAMS).
        self.stream = None
        if self.on_gpu:
            from . import ops
            self.stream = ops._side_stream(dev, self.main)
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        # The LAST bucket holds the first layers of the network, whose gradients arrive last: its pack + all-reduce + hand-back is the part of the
        # exchange nothing can hide (the backward pass is over).  Split its tail off into a small final bucket (PRN_EXCHANGE_TAIL_BYTES, default 2 MB: the stem
        # and the first blocks of stage 1), so that what runs after the last gradient is a 2 MB collective instead of a 25 MB one.
        tail = int(_os.environ.get("PRN_EXCHANGE_TAIL_BYTES", str(2 << 20)))
        if tail > 0 and self.buckets:
            last, keep, size = self.buckets[-1], [], 0
            while len(last) > 1 and size + last[-1].numel() * last[-1].element_size() <= tail:
                size += last[-1].numel() * last[-1].element_size()
                keep.insert(0, last.pop())
            if keep:
                self.buckets.append(keep)
        self.flat = [torch.empty(sum(p.numel() for p in b), device=dev, dtype=b[0].dtype) for b in self.buckets]
        # per-parameter windows of the flat buffers, built once: the pack / unpack are then ONE foreach copy each with no
        # per-step view construction (300 parameters: 3 ms of host time per step on the autograd thread otherwise)
        self.windows = [[v.view_as(p) for v, p in zip(f.split([p.numel() for p in b]), b)] for f, b in zip(self.flat, self.buckets)]
        self.where = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self.where[p] = bi
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.arrived = [0] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.next = 0
        # per parameter: how many ranks produced a gradient this step (all-reduced in finish(); optim.FusedAdam skips a tensor whose
        # count is 0, the way optim.Adam skips a parameter whose .grad is None -- reference train.py:362)
        self.index = {p: i for i, p in enumerate(self.params)}
        self.presence = torch.ones(len(self.params), device=dev, dtype=torch.float32)
        self._ones = torch.ones(len(self.params), device=dev, dtype=torch.float32)
        self._presence_work = None
        self._zeros = {}

    # called by autograd on the backward thread, once per parameter per backward
    def _on_grad(self, p):
        if _PROF is not None:
            import time
            t0 = time.perf_counter()
            self._on_grad_impl(p)
            _PROF["hooks"] = _PROF.get("hooks", 0.0) + time.perf_counter() - t0
            _PROF["n_hooks"] = _PROF.get("n_hooks", 0) + 1
            return
        self._on_grad_impl(p)

    def _on_grad_impl(self, p):
        bi = self.where[p]
        self.arrived[bi] += 1
        if self.arrived[bi] == len(self.buckets[bi]):
            self.ready[bi] = True
            self._launch_ready()

    def _launch_ready(self):
        # Collectives are issued strictly in bucket order, whatever order the gradients arrive in: every rank must post the
        # same sequence of all-reduces, and arrival order may differ between ranks (a parameter without gradient on one rank)
        while self.next < len(self.buckets) and self.ready[self.next]:
            self._launch(self.next)
            self.next += 1

    def _launch(self, bi):
        if _PROF is not None:
            import time
            t0 = time.perf_counter()
            self._launch_impl(bi)
            _PROF["launch"] = _PROF.get("launch", 0.0) + time.perf_counter() - t0
            return
        self._launch_impl(bi)

    def _launch_impl(self, bi):
        bucket, flat = self.buckets[bi], self.flat[bi]
        if self.on_gpu:
            from . import ops
            ops.wgrad_flush()                             # queued deferred weight gradients of this bucket must be in flight
        if self.on_gpu:
            # gradients of one bucket come from the backward's own stream AND (deferred weight gradients, ops.WGRAD_ASYNC)
            # from the side streams; the hook that completes a bucket may run under either
            cur = torch.cuda.current_stream()
            if cur.cuda_stream != self.stream.cuda_stream:
                self.stream.wait_stream(cur)
            if self.main.cuda_stream != cur.cuda_stream:
                self.stream.wait_stream(self.main)
            for st in ops.wgrad_streams():                 # (other side streams exist only with stream branches on)
                if st.cuda_stream != self.stream.cuda_stream:
                    self.stream.wait_stream(st)
            ctx = torch.cuda.stream(self.stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            torch._foreach_copy_(self.windows[bi], [p.grad for p in bucket])
            if self.avg_in_collective:
                work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                if self.world > 1:
                    flat.div_(self.world)
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((bi, work))

    def finish(self):
        """Block the compute stream until every bucket has been reduced and written back. Call after backward()."""
        if not self.active:
            return
        if _PROF is not None:
            import time
            t0 = time.perf_counter()
            self._finish_impl()
            _PROF["finish"] = _PROF.get("finish", 0.0) + time.perf_counter() - t0
            _PROF["steps"] = _PROF.get("steps", 0) + 1
            return
        self._finish_impl()

    def _finish_impl(self):
        # Buckets held back by a parameter that received no gradient on THIS rank: every rank must still post the same sequence
        # of collectives, so the missing gradient enters as zeros.  Whether ALL ranks lacked it (then the reference's optim.Adam would
        # skip the parameter: .grad is None, train.py:362) is exchanged as a per-parameter presence count that stays on the device:
        # optim.FusedAdam (its `exchange` attribute) leaves a tensor with count 0 untouched -- no host round trip.  Only `.grad` itself
        # differs from the single-process run (a zero tensor instead of None).
        missing = []
        for bi in range(self.next, len(self.buckets)):
            if any(p.grad is not None for p in self.buckets[bi]) or self.world > 1:
                for p in self.buckets[bi]:
                    if p.grad is None:
                        # (one zero tensor per such parameter, made once: nothing writes through `.grad` here -- the reduced values live in the bucket
                        # window `.grad` is pointed at below -- so it stays zero; a fresh torch.zeros_like per step was 43 fill launches on the compute stream)
                        z = self._zeros.get(p) if ALIAS_GRADS else None      # (with the copy-back variant the tensor receives the reduced mean: a fresh one each step)
                        if z is None or z.shape != p.shape or z.device != p.device:
                            z = torch.zeros_like(p)
                            if ALIAS_GRADS:
                                self._zeros[p] = z
                        p.grad = z
                        missing.append(self.index[p])
                self._launch(bi)
        self.next = len(self.buckets)
        # which parameters had a gradient on ANY rank: one more (tiny) all-reduce, posted by every rank in every step
        ctx = torch.cuda.stream(self.stream) if self.on_gpu else None
        if ctx is not None:
            ctx.__enter__()
        try:
            self.presence.copy_(self._ones)
            if missing:
                self.presence[torch.tensor(missing, device=self.presence.device)] = 0.0
            self._presence_work = dist.all_reduce(self.presence, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        for bi, work in self._pending:
            bucket, flat = self.buckets[bi], self.flat[bi]
            if self.on_gpu:
                with torch.cuda.stream(self.stream):
                    work.wait()                           # orders the SIDE stream (current here) after RCCL's completion
                    if not ALIAS_GRADS:
                        torch._foreach_copy_([p.grad for p in bucket], self.windows[bi])
            else:
                work.wait()
                if not ALIAS_GRADS:
                    torch._foreach_copy_([p.grad for p in bucket], self.windows[bi])
            if ALIAS_GRADS:
                # no copy back: the reduced values stay where RCCL left them and `.grad` becomes the parameter's window of the bucket
                # buffer (a second pass over all gradients saved; the windows are rewritten by the next step's pack, which the
                # exchange stream issues after waiting for the compute stream, i.e. after the optimizer has read them)
                for p, v in zip(bucket, self.windows[bi]):
                    p.grad = v
        if self._presence_work is not None:
            if self.on_gpu:
                with torch.cuda.stream(self.stream):
                    self._presence_work.wait()
            else:
                self._presence_work.wait()
            self._presence_work = None
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._pending.clear()
        self.arrived = [0] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.next = 0

    def remove(self):
        for h in self._handles:
            h.remove()


def all_reduce_mean_scalars(values, device):
    """Mean over ranks of a list of python floats / 0-d tensors (loss logging; non-finite flag)."""
    t = torch.stack([torch.as_tensor(v, dtype=torch.float64, device=device).reshape(()) for v in values])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
        t /= dist.get_world_size()
    return t
