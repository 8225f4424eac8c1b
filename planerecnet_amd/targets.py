"""GT-only preparation of the loss ON THE DEVICE -- the alternative to losses.TargetPrefetcher's worker processes.

What the reference does on the host per image and per plane (models/functions/losses.py:200-286: centre of mass, empty-mask flag and
the cv2 1/4 rescale of every GT mask; models/functions/vnl.py:43-70: np.flatnonzero of every plane region and three index draws per
region) is pixel work over ~15 MB of uint8 masks per batch of 8.  Here that work runs in four small HIP kernels
(csrc/prn_targets.hip) on the side stream, one batch ahead of the step that needs it; the host keeps only what is O(instances):

  submit(batch)   upload the packed masks -> prn_gt_mask_stats (pixels / sum x / sum y per region, per-segment counts + prefix),
                  prn_gt_quarter_masks -> the 56 x 24 bytes of region totals travel back into page-locked memory behind an event;
  get()           (a step later: the event has long fired) centre of mass and empty flags from the totals -> the reference's
                  centre-region loop over <= 8 instances x 4 levels (losses.PlaneRecNetLoss.assign_cells, shared with the host
                  path) -> cell lists / category maps uploaded, instance labels gathered from the 1/4 masks on the device;
                  per region n = int(0.3 * pixels) triplets -> prn_gt_sample_triplets maps RANKS to pixels.

Triplet ranks: `sampler="numpy"` draws them on the host from numpy's global stream exactly as vnl.py:43-55 does (choice + shuffle,
three times per region, regions in the reference's order) and injects them: the triplets are then BIT-IDENTICAL to the host path's
(parity tests).  `sampler="philox"` (default) draws them in the kernel from a counter-based generator: same distribution (uniform
with replacement per region; the reference's shuffle of i.i.d. draws changes nothing), different stream, no host work at all.

Exactness: pixel counts and coordinate sums are integers (exact); the centre of mass is float32(sum) / float32(count), which equals
the reference's float32 reductions whenever those are exact (sums < 2^24) and otherwise agrees to the last ulp except for torch's
summation-order rounding -- a centre would have to sit within an ulp of a grid-cell boundary for a target to differ.
"""
import collections
import ctypes
import os

import numpy as np
import torch

from . import ops
from .config import cfg
from .ops import _p, _stream, check, lib


class _Arena:
    """Small host arrays of one batch -> ONE page-locked buffer -> ONE asynchronous upload -> device views.  (A pin_memory() call per
    array costs ~0.1 ms of host time each, a pageable upload blocks the trainer until the side stream has caught up.)"""

    def __init__(self, nbytes=4 << 20):
        self.host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.np = self.host.numpy()
        self.free = None                                   # event after which the buffer may be rewritten

    def upload(self, arrays, device):
        """arrays: dict name -> numpy array (or None).  Returns dict name -> device tensor (same dtype / shape) or None."""
        if self.free is not None:
            self.free.synchronize()
        plan, off = {}, 0
        for k, a in arrays.items():
            if a is None:
                continue
            a = np.ascontiguousarray(a)
            n = a.nbytes
            if off + n + 16 > self.np.shape[0]:
                grow = torch.empty(max(2 * self.np.shape[0], off + n + 16), dtype=torch.uint8).pin_memory()
                grow.numpy()[:off] = self.np[:off]
                self.host, self.np = grow, grow.numpy()
            self.np[off:off + n] = a.reshape(-1).view(np.uint8)
            plan[k] = (off, n, a.dtype, a.shape)
            off += (n + 15) & ~15
        dev = torch.empty(max(off, 16), dtype=torch.uint8, device=device)
        dev[:off].copy_(self.host[:off], non_blocking=True)
        self.free = torch.cuda.Event()
        self.free.record()
        out = {k: None for k in arrays}
        for k, (o, n, dt, shp) in plan.items():
            out[k] = dev[o:o + n].view(_TORCH_DT[np.dtype(dt)]).view(shp) if n else torch.empty(shp, dtype=_TORCH_DT[np.dtype(dt)], device=device)
        return out


_TORCH_DT = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
             np.dtype(np.bool_): torch.bool, np.dtype(np.uint8): torch.uint8}


class DeviceTargetBuilder:
    """Drop-in for losses.TargetPrefetcher (submit / get / pending / discard / close); no worker processes."""

    def __init__(self, criterion, sampler=None, seed=0, depth=3, first_call=0):
        """seed: the Philox KEY of the device sampler -- (run seed << 32) | rank, so that ranks and runs draw different streams;
        first_call: the batch COUNTER the stream starts at -- the global iteration a resumed run continues from."""
        self.criterion = criterion
        self.sampler = sampler or os.environ.get("PRN_TARGET_SAMPLER", "philox")
        if self.sampler not in ("philox", "numpy"):
            raise ValueError("sampler must be 'philox' or 'numpy'")
        self.seed, self.calls = int(seed), int(first_call)
        self.queue = collections.deque()
        self._stage = []                                   # rotating page-locked staging buffers for the packed masks
        self._depth = depth
        self._arenas, self._arena_i = [_Arena() for _ in range(2 * (depth + 1))], 0
        self._totals = [torch.empty(4096, dtype=torch.int64).pin_memory() for _ in range(depth + 2)]
        self._totals_i = 0
        self.nworkers = 0
        self.host_ms = 0.0                                 # host WALL time spent in submit + get (includes waiting for the device when the host runs ahead)
        self.host_cpu_ms = 0.0                             # CPU time of the calling thread inside submit + get: the host WORK of the preparation

    # ------------------------------------------------------------------------------------------------------------------
    def _staging(self, nbytes):
        for st in self._stage:
            if st["free"] is None or st["free"].query():
                if st["buf"].numel() < nbytes:
                    st["buf"] = torch.empty(int(nbytes * 1.25), dtype=torch.uint8).pin_memory()
                return st
        if len(self._stage) < self._depth + 1:
            st = {"buf": torch.empty(int(nbytes * 1.25), dtype=torch.uint8).pin_memory(), "free": None}
            self._stage.append(st)
            return st
        self._stage[0]["free"].synchronize()               # (more batches in flight than buffers: wait for the oldest upload)
        return self._stage[0]

    def _arena(self):
        self._arena_i = (self._arena_i + 1) % len(self._arenas)
        return self._arenas[self._arena_i]

    @property
    def pending(self):
        return self.queue[0] if self.queue else None

    @torch.no_grad()
    def submit(self, gt_instances, hw, mask_feat_size=None):
        import time
        t0, c0 = time.perf_counter(), time.thread_time()
        H, W = hw
        dev = torch.device("cuda", torch.cuda.current_device())
        B = len(gt_instances)
        N_per = [int(g["masks"].shape[0]) for g in gt_instances]
        Ntot = sum(N_per)
        img_first = np.concatenate([[0], np.cumsum(N_per)]).astype(np.int32)
        nbytes = Ntot * H * W
        st = self._staging(max(nbytes, 16))
        off = 0
        for g in gt_instances:                             # pack the uint8 masks of all images back to back (page-locked)
            m = g["masks"]
            if m.dtype != torch.uint8:
                m = m.to(torch.uint8)
            n = m.numel()
            if n:
                st["buf"][off:off + n].copy_(m.reshape(-1))
            off += n
        main = torch.cuda.current_stream()
        side = ops._side_stream(dev, main)
        side.wait_stream(main)
        R, nseg = Ntot + B, int(lib.prn_gt_segments(H, W))
        with torch.cuda.stream(side):
            masks = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            masks[:nbytes].copy_(st["buf"][:nbytes], non_blocking=True)
            st["free"] = torch.cuda.Event()
            st["free"].record()
            first_d = self._arena().upload({"first": img_first}, dev)["first"]
            segcnt = torch.empty(R * nseg, dtype=torch.uint8, device=dev)
            segstart = torch.empty(R * (nseg + 1), dtype=torch.int32, device=dev)
            totals = torch.empty(R * 3, dtype=torch.int64, device=dev)
            check(lib.prn_gt_mask_stats(_p(masks), _p(first_d), B, Ntot, H, W, _p(segcnt), _p(segstart), _p(totals), _stream()), "prn_gt_mask_stats")
            small = torch.empty(Ntot, H // 4, W // 4, dtype=torch.uint8, device=dev)
            if Ntot:                                           # (a batch without a single plane: an empty tensor has no address to pass)
                check(lib.prn_gt_quarter_masks(_p(masks), _p(small), Ntot, H, W, _stream()), "prn_gt_quarter_masks")
            self._totals_i = (self._totals_i + 1) % len(self._totals)
            if self._totals[self._totals_i].numel() < R * 3:
                self._totals[self._totals_i] = torch.empty(R * 3 * 2, dtype=torch.int64).pin_memory()
            totals_h = self._totals[self._totals_i][:R * 3]
            totals_h.copy_(totals, non_blocking=True)
            done = torch.cuda.Event(blocking=True)          # (a waiting host thread sleeps instead of spinning: the wait is not host WORK)
            done.record()
        small_meta = [{k: (g[k].cpu() if torch.is_tensor(g[k]) else g[k]) for k in ("boxes", "classes", "plane_paras", "k_matrix")} for g in gt_instances]
        self.queue.append({"hw": (H, W), "feat": mask_feat_size, "B": B, "N_per": N_per, "Ntot": Ntot, "img_first": img_first, "first_d": first_d,
                           "masks": masks, "segstart": segstart, "segcnt": segcnt, "small": small, "totals_h": totals_h, "done": done, "meta": small_meta})
        self.host_ms += (time.perf_counter() - t0) * 1e3
        self.host_cpu_ms += (time.thread_time() - c0) * 1e3

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _finish(self, job, gt_depths, device):
        """Host: O(instances) bookkeeping from the region totals; device: label gather, triplet sampling, the loss's own uploads."""
        crit = self.criterion
        H, W = job["hw"]
        fh, fw = job["feat"] if job["feat"] is not None else (H // 4, W // 4)
        B, N_per, Ntot, first = job["B"], job["N_per"], job["Ntot"], job["img_first"]
        job["done"].synchronize()                          # recorded a step ago: no wait in steady state
        tot = job["totals_h"].view(-1, 3)
        cnt_t, sx_t, sy_t = tot[:, 0], tot[:, 1], tot[:, 2]
        m00 = cnt_t.to(torch.float32).clamp(min=1e-6)
        cx_all, cy_all = sx_t.to(torch.float32) / m00, sy_t.to(torch.float32) / m00
        L = len(crit.num_grids)
        level_start = np.concatenate([[0], np.cumsum([g * g for g in crit.num_grids])])
        cell_ids, which_all, cate_rows, n_pos, num_ins = [], [], [[] for _ in range(L)], [], 0
        # the float arithmetic of the assignment for all instances of the batch at once (losses.cell_regions); the per-image loop only places cells
        regions = crit.cell_regions(np.concatenate([np.asarray(m["boxes"], dtype=np.float64).reshape(-1, 4) for m in job["meta"]]) if Ntot else np.zeros((0, 4)),
                                    cx_all[:Ntot], cy_all[:Ntot], (fh, fw))
        nonempty_all = (cnt_t > 0).numpy()
        for b in range(B):
            lo, hi = int(first[b]), int(first[b + 1])
            meta = job["meta"][b]
            which_l, cate_l, ind_l, order_l = crit.assign_cells(None, meta["classes"], None, None, nonempty_all[lo:hi], (fh, fw), regions, lo)
            cell_ids.append(np.concatenate([level_start[lv] + np.asarray(order_l[lv], dtype=np.int64) for lv in range(L)]))
            which_all.append(np.concatenate([np.asarray(w, dtype=np.int64) for w in which_l]) + lo)
            n_pos.append(int(cell_ids[-1].shape[0]))
            num_ins += sum(int(i.sum()) for i in ind_l)
            for lv in range(L):
                cate_rows[lv].append(cate_l[lv].flatten())
        n_cells = int(level_start[-1])
        cell_gidx = np.concatenate([b * n_cells + cell_ids[b] for b in range(B)]) if B else np.zeros(0, np.int64)
        cell_u, cell_inv, cell_cnt = np.unique(cell_gidx, return_inverse=True, return_counts=True)
        cell_mult = int(cell_cnt.max()) if cell_cnt.size else 1
        vnl_np, vnl_meta = self._vnl_host(job, cnt_t) if cfg.use_plane_loss else ({}, None)
        arrays = {"which": np.concatenate(which_all) if which_all else np.zeros(0, np.int64),
                  "n_pos_f": np.asarray(n_pos, dtype=np.float32),
                  "cell_gidx": cell_u if cell_mult > 1 else cell_gidx,
                  "cell_inv": cell_inv.astype(np.int64) if cell_mult > 1 else None,
                  "cell_ids": np.concatenate(cell_ids) if cell_ids else np.zeros(0, np.int64),
                  "pos_img": np.repeat(np.arange(B), n_pos),
                  # level-major, image-minor flattening == the reference's cat order (losses.py:121-131)
                  "cate_labels": torch.cat([r for lv in range(L) for r in cate_rows[lv]]).numpy()}
        arrays.update({"vnl_" + k: v for k, v in vnl_np.items()})
        d = self._arena().upload(arrays, device)
        small = job["small"]
        if small.shape[1] != fh or small.shape[2] != fw:     # (mask features of another size than H/4 x W/4: place like losses.py:268-270)
            pad = torch.zeros(small.shape[0], fh, fw, dtype=torch.uint8, device=device)
            pad[:, :min(fh, small.shape[1]), :min(fw, small.shape[2])] = small[:, :fh, :fw]
            small = pad
        h = {"B": B, "hw": (H, W), "feat": (fh, fw), "n_pos": n_pos, "num_ins": num_ins, "n_pos_f": d["n_pos_f"],
             "cell_gidx": d["cell_gidx"], "cell_inv": d["cell_inv"], "cells_unique": cell_mult <= 2,
             "cell_ids": d["cell_ids"], "pos_img": d["pos_img"], "ins_labels": small.index_select(0, d["which"]),
             "cate_labels": d["cate_labels"],
             "vnl": self._vnl_device(job, vnl_meta, {k[4:]: v for k, v in d.items() if k.startswith("vnl_")}, device) if cfg.use_plane_loss else None}
        return crit.upload(h, gt_depths, device)

    def _vnl_host(self, job, cnt_t):
        """Segment bookkeeping of vnl.py:119-140 from the region pixel counts -> (arrays to upload, host-side meta)."""
        H, W = job["hw"]
        B, N_per, Ntot, first = job["B"], job["N_per"], job["Ntot"], job["img_first"]
        ratio = self.criterion.vnl.sample_ratio
        cnt = cnt_t.numpy()
        seg_len, seg_img, seg_plane, seg_region, normals, fx, fy, ranks = [], [], [], [], [], [], [], []
        for b in range(B):
            meta = job["meta"][b]
            K = meta["k_matrix"].numpy()
            fx.append(K[0, 0]); fy.append(K[1, 1])
            planes = meta["plane_paras"].numpy()[:, :3]
            regions = [(int(first[b]) + i, True, planes[i]) for i in range(N_per[b])]
            if int(cnt[Ntot + b]) > 0:                       # the pixels no plane covers (vnl.py:127-131)
                regions.append((Ntot + b, False, np.zeros(3)))
            for r, is_plane, nrm in regions:
                num = int(cnt[r])
                if not num <= W * H:
                    raise AssertionError()
                n = int(num * ratio)
                if self.sampler == "numpy":                  # vnl.py:43-55: three (choice, shuffle) pairs from numpy's global stream
                    trio = []
                    for _ in range(3):
                        p = np.random.choice(num, n, replace=True)
                        np.random.shuffle(p)
                        trio.append(p)
                    ranks.append(np.stack(trio, 0))
                seg_len.append(n); seg_img.append(b); seg_plane.append(is_plane); seg_region.append(r); normals.append(nrm)
        seg_len = np.asarray(seg_len, dtype=np.int64)
        n_seg, n_tot = len(seg_len), int(seg_len.sum())
        arrays = {"seg_len": seg_len, "seg_region": np.asarray(seg_region, np.int32), "seg_img32": np.asarray(seg_img, np.int32),
                  "N": np.asarray(N_per, dtype=np.float64), "fx": np.asarray(fx, dtype=np.float64), "fy": np.asarray(fy, dtype=np.float64),
                  "seg_start": (np.concatenate([[0], np.cumsum(seg_len)[:-1]]) if n_seg else np.zeros(0, np.int64)).astype(np.int64),
                  "seg_img": np.asarray(seg_img, dtype=np.int64), "seg_is_plane": np.asarray(seg_plane, dtype=np.bool_),
                  "seg_normal": np.asarray(normals, dtype=np.float64).reshape(-1, 3),
                  "ranks": np.concatenate(ranks, 1).astype(np.int32) if (self.sampler == "numpy" and n_tot) else None}
        return arrays, {"n_seg": n_seg, "n_tot": n_tot}

    def _vnl_device(self, job, meta, d, device):
        H, W = job["hw"]
        B, Ntot = job["B"], job["Ntot"]
        n_seg, n_tot = meta["n_seg"], meta["n_tot"]
        seg = torch.repeat_interleave(torch.arange(n_seg, device=device, dtype=torch.int32), d["seg_len"], output_size=n_tot) if n_tot else \
            torch.zeros(0, dtype=torch.int32, device=device)
        gid = torch.empty(3, n_tot, dtype=torch.int32, device=device)
        if n_tot:
            check(lib.prn_gt_sample_triplets(_p(job["masks"]), _p(job["first_d"]), B, Ntot, H, W, _p(job["segstart"]), _p(seg), _p(d["seg_region"]), _p(d["seg_img32"]),
                                             _p(d["ranks"]), ctypes.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF), ctypes.c_uint64(self.calls), n_tot, _p(gid), _stream()), "prn_gt_sample_triplets")
        return {"B": B, "n_seg": n_seg, "n_tot": n_tot, "npts": B * H * W, "N": d["N"], "fx": d["fx"], "fy": d["fy"], "gid": gid, "seg": seg,
                "seg_start": d["seg_start"], "seg_img": d["seg_img"], "seg_is_plane": d["seg_is_plane"], "seg_normal": d["seg_normal"]}

    def get(self, gt_depths, device, overlap=False):
        """Targets of the OLDEST submitted batch.  overlap: issue the device work on the weight-gradient side stream (idle during
        the forward pass); the loss waits for `ready`."""
        import time
        t0, c0 = time.perf_counter(), time.thread_time()
        job = self.queue.popleft()
        self.calls += 1                                    # (also the stream position of the device sampler)
        main = torch.cuda.current_stream()
        side = ops._side_stream(torch.device(device), main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            t = self._finish(job, gt_depths, device)
            if overlap:
                t.ready = torch.cuda.Event()
                t.ready.record()
        if not overlap:
            main.wait_stream(side)
        self.host_ms += (time.perf_counter() - t0) * 1e3
        self.host_cpu_ms += (time.thread_time() - c0) * 1e3
        return t

    def discard(self):
        while self.queue:
            self.queue.popleft()["done"].synchronize()

    def close(self):
        self.discard()
