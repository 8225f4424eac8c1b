"""Adam as ONE device launch over all parameter tensors (include/prn.h: prn_adam_step) -- the optimizer of the reference's
train.py:251-256 (optim.Adam with per-group learning rates; no weight decay, no amsgrad) behind torch.optim's interface:
`param_groups` (so `set_lr` keeps working), `zero_grad`, `state_dict` with the usual exp_avg / exp_avg_sq / step entries, and
the `found_inf` / `grad_scale` attributes torch's fused optimizers take from a GradScaler (train.py uses `found_inf` to skip
the update on a non-finite loss without reading the loss on the host)."""
import ctypes

import torch

from ._lib import check, lib


class FusedAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.found_inf = None
        self.grad_scale = None
        self.exchange = None          # a parallel.GradAllReduce: tensors for which no rank produced a gradient are skipped (see step)
        self._tab = None

    def load_state_dict(self, state_dict):
        """torch re-creates the state tensors (new exp_avg / exp_avg_sq storage, `step` possibly left on the host): the launch
        tables hold raw pointers, so they are rebuilt at the next step."""
        super().load_state_dict(state_dict)
        self._tab = None

    # ---- tables of the launch: rebuilt when the set of parameters that have a gradient changes, when a parameter's storage moved
    # (module.to() / .float(), a re-linked .data) or after load_state_dict -- the kernel works through raw pointers
    def _build(self, plist):
        dev = plist[0][0].device
        ce = lib.prn_adam_chunk_elems()
        chunks, numel, old_steps = [], [], []
        for i, (p, _) in enumerate(plist):
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise RuntimeError("FusedAdam: parameters must be contiguous fp32 device tensors")
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
            for k in ("exp_avg", "exp_avg_sq"):
                if st[k].device != p.device or st[k].dtype != torch.float32 or not st[k].is_contiguous():
                    st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
            old_steps.append(st.get("step"))
            numel.append(p.numel())
            chunks += [(i, o) for o in range(0, p.numel(), ce)]
        # per-tensor update counters (optim.Adam's state['step']) in ONE device array, each state entry a 0-d view of it; values of a
        # loaded / previous state are carried over without a device-to-host read (a loaded state may keep them on the host)
        step = torch.stack([torch.zeros((), dtype=torch.float32, device=dev) if t is None else
                            torch.as_tensor(t, dtype=torch.float32).reshape(()).to(dev, non_blocking=True) for t in old_steps])
        for i, (p, _) in enumerate(plist):
            self.state[p]["step"] = step[i]
        ptr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)      # noqa: E731
        tab = {"ids": [id(p) for p, _ in plist], "pptr": [p.data_ptr() for p, _ in plist], "dev": dev, "step": step, "n": len(plist), "nchunks": len(chunks),
               "chunks": torch.tensor(chunks, dtype=torch.int32, device=dev), "numel": torch.tensor(numel, dtype=torch.int32, device=dev),
               "p": ptr([p for p, _ in plist]), "m": ptr([self.state[p]["exp_avg"] for p, _ in plist]),
               "v": ptr([self.state[p]["exp_avg_sq"] for p, _ in plist]),
               "g": torch.empty(len(plist), dtype=torch.int64, device=dev), "lr": torch.empty(len(plist), dtype=torch.float32, device=dev),
               "lr_key": None, "slot": 0,
               # The gradient pointers (and the learning rates, when they change) travel through page-locked staging buffers; the
               # host runs up to a step ahead of the GPU, so a buffer is only rewritten after the copy that read it has executed.
               "ring": [{"g": torch.empty(len(plist), dtype=torch.int64).pin_memory(), "lr": torch.empty(len(plist), dtype=torch.float32).pin_memory(),
                         "done": None} for _ in range(4)]}
        for r in tab["ring"]:
            r["g_np"], r["lr_np"] = r["g"].numpy(), r["lr"].numpy()
        return tab

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad walks every parameter through its foreach / profiler machinery (~0.6 ms per step for the 477 tensors of
        PlaneRecNet_101); dropping the gradients is all the training loop asks for."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        from . import ops
        if ops.WGRAD_ASYNC:
            # deferred weight gradients (ops.set_wgrad_async) are written to `.grad` by the side stream's launches: a loop that forgot ops.wgrad_join() would
            # step without them.  Idempotent -- nothing queued, nothing launched; the stream wait is two runtime calls.
            ops.wgrad_join()
        # one pass over the parameters: (parameter, group, gradient) of those that have a gradient
        plist, grads = [], []
        for g in self.param_groups:
            for p in g["params"]:
                gr = p.grad
                if gr is not None:
                    plist.append((p, g))
                    grads.append(gr)
        if not plist:
            return loss
        betas, eps = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        if any(g["betas"] != betas or g["eps"] != eps for g in self.param_groups):
            raise RuntimeError("FusedAdam: betas / eps must be the same in every parameter group")
        tab = self._tab
        if tab is None or tab["ids"] != [id(p) for p, _ in plist] or tab["pptr"] != [p.data_ptr() for p, _ in plist]:
            tab = self._tab = self._build(plist)
        r = tab["ring"][tab["slot"]]
        tab["slot"] = (tab["slot"] + 1) % len(tab["ring"])
        if r["done"] is not None:
            r["done"].synchronize()
        gptr = []
        for gr in grads:
            if gr.dtype != torch.float32 or not gr.is_contiguous():
                raise RuntimeError("FusedAdam: gradients must be contiguous fp32 tensors")
            gptr.append(gr.data_ptr())
        r["g_np"][:] = gptr
        tab["g"].copy_(r["g"], non_blocking=True)
        lr_key = tuple(float(g["lr"]) for g in self.param_groups)
        if tab["lr_key"] != lr_key:
            r["lr_np"][:] = [float(g["lr"]) for _, g in plist]
            tab["lr"].copy_(r["lr"], non_blocking=True)
            tab["lr_key"] = lr_key
        r["done"] = torch.cuda.Event()
        r["done"].record()
        vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)      # noqa: E731
        fi = self.found_inf.float().reshape(()) if self.found_inf is not None else None
        gs = self.grad_scale.float().reshape(()) if self.grad_scale is not None else None
        stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(tab["dev"].index))
        present = pidx = None
        ex = self.exchange
        if ex is not None and getattr(ex, "active", False) and ex.presence is not None:
            # data-parallel: the exchange zero-fills gradients a rank did not produce (every rank must post the same collectives), so
            # "no rank had a gradient" arrives as an all-reduced count on the device instead of a host-side None; the kernel leaves such
            # tensors alone like optim.Adam leaves a parameter whose .grad is None (reference train.py:362)
            if tab.get("pidx_for") is not ex:
                tab["pidx"] = torch.tensor([ex.index[p] for p, _ in plist], dtype=torch.int32, device=tab["dev"])
                tab["pidx_for"] = ex
            present, pidx = ex.presence, tab["pidx"]
        check(lib.prn_adam_step_masked(vp(tab["chunks"]), tab["nchunks"], tab["n"], vp(tab["p"]), vp(tab["g"]), vp(tab["m"]), vp(tab["v"]), vp(tab["numel"]),
                                       vp(tab["lr"]), vp(tab["step"]), vp(fi), vp(gs), float(betas[0]), float(betas[1]), float(eps), vp(present), vp(pidx),
                                       stream), "prn_adam_step")
        # the kernel wrote through raw pointers: advance the version counters like an in-place torch op would -- the flipped /
        # Winograd-domain / BatchNorm-folded weight caches of this package are keyed on them
        torch.autograd.graph.increment_version([p for p, _ in plist])
        return loss
