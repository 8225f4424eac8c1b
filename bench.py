#!/usr/bin/env python
"""Headline benchmark: images/s of one PlaneRecNet_101 optimisation step (forward + joint loss + backward +
gradient exchange + Adam) at 480x640, per-GPU batch 8, synthetic data, random-init weights (BASELINE.json).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` = global images / max-over-ranks wall time of exactly K steps bracketed by
barrier + device synchronize.  Inputs (images, GT) are resident in HBM / host memory before the timed region.

`--workload` selects the BASELINE.json configuration (default c3, the one the metric is quoted on):
  c3  PlaneRecNet_101 train step, per-GPU batch 8, 480x640                      (configs[2]; N > 1 is configs[3])
  c1  PlaneRecNet_50  eval forward + post-process, batch 1, 480x640             (configs[0]: the reference's CPU-runnable case;
                                                                                  the GPU runs it, cpu_baseline is the CPU side)
  c2  PlaneRecNet_50  eval forward + post-process, batch 8, 480x640             (configs[1])
  c5  PlaneRecNet_101 eval forward + post-process, batch 4, max_size=960 -> 736x960   (configs[4])
For the inference workloads a step is one batch through net.eval() (BatchNorm folded into the conv epilogues, on-device
post-process, list[dict] result) and `value` is images / s.

Extra objects:
  roofline     -- the dominant MFMA kernel family BY TIME of the bracketed step, whichever that is (since round 3: the weight-gradient
                  GEMMs, conv_wgrad_kernel; `roofline.kernel` names it, `conv_kernel_groups` has every group).  Every launch of
                  one extra, untimed step is bracketed by HIP events on the launch stream; achieved = sum of ALGORITHMIC fp32
                  FLOPs (2*M*N*K of the GEMMs the launches evaluate) / sum of event durations.  peak = 157.3 TFLOP/s (fp32 MFMA,
                  MI355X_MICROARCH.md); launches on the 16-bit pipe (split_gemm_kernel: forward / input-gradient plain GEMMs; wgrad16_kernel: their
                  weight gradients) are ALSO priced against the pipe they run on, in `roofline.split_gemm` / `roofline.wgrad16`.  When such a
                  family is the dominant one, the headline is against the roofline that BINDS it: the larger of (executed piece products /
                  16-bit pipe peak) and (algorithmic bytes / 8 TB/s) per launch -- for the short-K 1x1 layers the memory side; `fp32_equivalent`
                  (algorithmic FLOPs over the fp32 MFMA peak) and `executed_*` sit next to it.  `traffic` puts
                  the PMC bytes per launch (profiles/*pmc_traffic.json) next to the algorithmic bytes per launch logged live
                  (4 B x operand + result elements of every launch).  `kernels` lists the other families the same way.
  cpu_baseline -- the oracle (CPU restatement proven equal to the reference) timed on the host cores on a bounded sample:
                  ONE image of the same workload; 1 warm-up + 3 timed iterations, median.  kind = "port".
  fp32_only_run -- the same step re-timed with every contraction on the fp32 MFMA kernels (prn_gemm_opts.split_mode = 0, what
                  PRN_SPLIT_GEMM=0 selects): the default sends the large plain GEMMs through the 16-bit matrix pipe as fp16-piece products
                  (`dtype` says so); this is the step's rate in the reference's arithmetic class, in the same process on the same board.
  conditioned_run (c1 / c2 / c5) -- the same batch re-timed after shifting inst_head.cate_pred.bias until every image keeps >= 5 detections
                  through matrix NMS: at the init state the post-process runs on (almost) empty candidate sets.
  dcn_offsets_run (c3) -- the same step re-timed after giving the DCN offset / modulator convs non-zero weights (~0.6 px r.m.s.
                  offsets): the headline runs at the reference's init state (offset convs zero, models/dcn.py:32-43), where
                  every deformable gather is a regular 3x3 pattern -- the best case for locality.

Diagnostics (environment, stderr only; none of them changes the timed work except FIXED_TARGETS):
  PRN_BENCH_PHASES=1        host time per phase of a step (get / submit / fwd / loss / bwd / adam) and GPU time per step
  PRN_BENCH_GAP=1           GPU time between the end of Adam and the next forward's first kernel, and the compute stream's wait at wgrad_join
  PRN_BENCH_FIXED_TARGETS=1 the same loss targets every step (nothing fetched from the prefetch workers): isolates the boundary work
  PRN_FORCE_EXCHANGE=1      the gradient exchange (hooks, bucket pack, RCCL all-reduce) with one rank in the HEADLINE timing (the default
                            run reports the same probe as `exchange_probe.exchange_overhead_ms` from extra steps)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# Hardware queues the HIP runtime multiplexes this process's streams onto (compute stream, weight-gradient / exchange side stream, RCCL's
# own streams with N > 1).  3 is the measured optimum with ONE rank; with N > 1 it is unmeasured, hence an explicit
# knob: `--hw-queues K` (or GPU_MAX_HW_QUEUES in the environment) -- read here because it must be set before the runtime initialises.
for _i, _a in enumerate(sys.argv):
    if _a == "--hw-queues" and _i + 1 < len(sys.argv):
        os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[_i + 1]
    elif _a.startswith("--hw-queues="):
        os.environ["GPU_MAX_HW_QUEUES"] = _a.split("=", 1)[1]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "img/s fwd+bwd @480×640 ResNet101-DCN, 1/2/4/8 MI355X + roofline %"
PEAK_FP32_MFMA_TFLOPS = 157.3
# dense bf16 MFMA peak: 256 CUs x 4 SIMDs x 1024 FLOP/clk (v_mfma_f32_32x32x16_bf16: 32768 FLOPs in 8 passes of 4 clk) x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 256 * 4 * 1024 * 2.4e9 / 1e12


PIPE16 = ("split_gemm_kernel", "wgrad16_kernel")      # profiler families whose launches issue 16-bit piece products (work credited = products x 2*M*N*K)


def mfma_peak(kernel):
    """Peak of the matrix pipe a kernel family issues on: the split GEMM runs bf16 piece products, everything else fp32 MFMA."""
    return PEAK_BF16_MFMA_TFLOPS if kernel in PIPE16 else PEAK_FP32_MFMA_TFLOPS
PEAK_HBM_GBS = 8000.0


def synth_batch(B, H, W, seed, device):
    """SURVEY.md 8(d) synthetic batch (same generator as the parity tests; oracle/synth.py is test infrastructure, so the
    generator is restated here for the product path)."""
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g)
    depths = 0.5 + 4.0 * torch.rand(B, 1, H, W, generator=g)
    inst = []
    for _ in range(B):
        n = int(rng.randint(3, 9))
        masks = np.zeros((n, H, W), np.uint8)
        boxes = np.zeros((n, 4), np.float64)
        for i in range(n):
            bw, bh = int(rng.randint(max(W // 16, 8), W // 2)), int(rng.randint(max(H // 16, 8), H // 2))
            x0, y0 = int(rng.randint(0, W - bw)), int(rng.randint(0, H - bh))
            masks[i, y0:y0 + bh, x0:x0 + bw] = 1
            boxes[i] = (x0, y0, x0 + bw, y0 + bh)
        nrm = rng.randn(n, 3)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        paras = np.concatenate([nrm, rng.rand(n, 1) * 3.0, np.zeros((n, 2))], 1)
        K = np.array([[577.0, 0, W / 2], [0, 577.0, H / 2], [0, 0, 1]], np.float64)
        inst.append({"masks": torch.from_numpy(masks), "boxes": torch.from_numpy(boxes), "classes": torch.zeros(n, dtype=torch.int64),
                     "plane_paras": torch.from_numpy(paras), "k_matrix": torch.from_numpy(K)})
    # annotations stay on the host (the loss's GT-only preparation runs there, in worker processes: shared memory, so a
    # batch is handed over as handles the way a DataLoader worker's batch is); images and GT depth live in HBM
    try:
        for g_ in inst:
            for v in g_.values():
                v.share_memory_()
    except RuntimeError:                                     # no shared-memory segment available: the batch is pickled by value
        pass
    return images.to(device), inst, depths.to(device)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the latest committed PMC summary (profiles/*pmc_traffic.json, produced by
    tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction). PMC counters cannot be
    read inside this process, so this is the profiled value of the same command, or None when no summary exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    fams = json.load(open(files[-1])).get("families", {})
    fam = fams.get(kernel)
    if fam is None and kernel == "split_gemm_kernel":           # (the profiler family of both piece formats; the fp16 one's kernel is split16_gemm_kernel)
        fam = fams.get("split16_gemm_kernel")
    if fam is None and kernel == "dcnv2_fwd_kernel":            # (round 5: the windowed forward is dcnv2_fwd2_kernel; the PMC summary keys it by the common prefix)
        fam = fams.get("dcnv2_fwd")
    if fam is not None and "hbm_bytes_per_launch" not in fam:
        fam = None
    return None if fam is None else {"bytes_per_launch": fam["hbm_bytes_per_launch"], "source": os.path.relpath(files[-1], ROOT)}


def physical_cores():
    """Physical core count of the host (unique (package, core) pairs of /proc/cpuinfo; falls back to the logical count)."""
    try:
        seen, pkg = set(), "0"
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                seen.add((pkg, ln.split(":")[1].strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def _cpu_sample(config_name, H, W, train, B, threads, n_timed, budget_s):
    """One setting of `cpu_baseline`: B images per iteration on `threads` threads, 1 warm-up + `n_timed` timed iterations (fewer when the
    warm-up shows they would not fit `budget_s`)."""
    from oracle import loss_ref, model_ref, synth
    arch = model_ref.ARCH[config_name]
    sd = synth.make_state_dict(config_name, seed=0)
    if train:
        sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
        leaves = [v for v in sd.values() if v.requires_grad]
    torch.set_num_threads(threads)
    x, inst, gtd = synth.make_batch(B, H, W, seed=0)
    times = []
    t_start = time.perf_counter()
    for it in range(1 + n_timed):
        np.random.seed(0)
        t0 = time.perf_counter()
        if train:
            out = model_ref.forward(sd, x, arch, training=True)
            ls = loss_ref.joint_loss(*out, inst, gtd)
            torch.autograd.grad(sum(ls.values()).sum(), leaves, allow_unused=True)
        else:
            with torch.no_grad():
                model_ref.inference(sd, x, arch)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start + times[-1] > budget_s and it < n_timed:
            break                                              # (the next iteration would not fit the budget)
    med = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {"images": B, "threads": threads, "img_per_s": B / med, "s_per_iteration": [round(t, 2) for t in times], "warm_up_only": len(times) == 1}


def cpu_baseline(config_name, H, W, train, batch=1, budget_s=30.0):
    """The oracle (CPU restatement of the reference, oracle/) on the host cores of this box, on a BOUNDED sample of the workload (BASELINE.md 4 / SURVEY.md 8d):
    fwd + loss + bwd (train) or eval forward + post-process.  Two settings are timed and the faster one is the reported `value` (both are in `samples`):
    one image on min(16, physical cores) threads -- on the 256-core boxes a full-width OpenMP team is ~500x SLOWER on this model (fork / join on hundreds of
    small ops; 64 threads at batch 8 did not finish a warm-up iteration in ten minutes) -- and the workload's own batch on the same team.
    Each setting runs in a CHILD process of this script (`--cpu-sample`, no GPU work) under a hard wall-clock limit, so that a host on which the
    oracle crawls costs the bench a bounded time and is reported as `timed_out` instead of stalling the run; 1 warm-up + 3 (batch: 2) timed iterations, median."""
    import subprocess
    phys = physical_cores()
    threads = max(1, min(16, phys))
    settings = [(1, threads, 3, budget_s)] + ([(batch, threads, 2, budget_s)] if batch > 1 else [])
    samples = []
    for B, th, n_timed, budget in settings:
        spec = json.dumps({"config": config_name, "H": H, "W": W, "train": bool(train), "B": B, "threads": th, "n_timed": n_timed, "budget_s": budget})
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(th))
        limit = 3.0 * budget + 60.0                            # (the child imports torch and builds the weights first: ~10 s here, 1-2 min on a cold image)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-sample", spec], env=env, capture_output=True, text=True, timeout=limit)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                raise RuntimeError("cpu sample failed (exit %d): %s" % (r.returncode, r.stderr[-400:]))
            samples.append(json.loads(line[-1]))
        except subprocess.TimeoutExpired:                      # (subprocess.run has killed the child it started)
            samples.append({"images": B, "threads": th, "img_per_s": None, "timed_out_after_s": limit})
    done = [r for r in samples if r["img_per_s"]]
    if not done:
        return {"value": None, "unit": "img/s", "cores": threads, "physical_cores": phys, "kind": "port", "sample": "no setting finished within its limit", "samples": samples}
    best = max(done, key=lambda r: r["img_per_s"])
    return {"value": best["img_per_s"], "unit": "img/s", "cores": best["threads"], "physical_cores": phys, "kind": "port",
            "sample": "%d image(s) per iteration, %s %s at %dx%d, torch CPU fp32 oracle on %d threads, 1 warm-up + %d timed iterations (median; s per iteration: %s)"
                      % (best["images"], config_name, "fwd+loss+bwd" if train else "eval forward + post-process", H, W, best["threads"],
                         len(best["s_per_iteration"]) - 1, ", ".join("%.2f" % t for t in best["s_per_iteration"])),
            "samples": samples}


WORKLOADS = {   # name -> (config, per-GPU batch, H, W, train?, description)
    "c3": ("PlaneRecNet_101_config", 8, 480, 640, True, "train step (fwd + 5-term loss + bwd + grad all-reduce + Adam)"),
    "c1": ("PlaneRecNet_50_config", 1, 480, 640, False, "eval forward + on-device post-process"),
    "c2": ("PlaneRecNet_50_config", 8, 480, 640, False, "eval forward + on-device post-process"),
    "c5": ("PlaneRecNet_101_config", 4, 736, 960, False, "eval forward + on-device post-process, max_size=960"),
}


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-sample":   # (the child of cpu_baseline: one setting, one JSON line, no GPU)
        q = json.loads(sys.argv[2])
        print(json.dumps(_cpu_sample(q["config"], q["H"], q["W"], q["train"], q["B"], q["threads"], q["n_timed"], q["budget_s"])), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS), help="BASELINE.json configuration (c3 = headline train step)")
    ap.add_argument("--config", default=None, help="override the workload's config")
    ap.add_argument("--batch", type=int, default=None, help="override the workload's per-GPU batch")
    ap.add_argument("--dcn-offsets", type=float, default=0.6, help="r.m.s. offset (pixels) of the extra `dcn_offsets_run` (0: skip it)")
    ap.add_argument("--sync-wgrad", action="store_true", help="weight gradients in line with the input-gradient chain (A/B of ops.WGRAD_ASYNC)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--targets", choices=("device", "workers"), default=os.environ.get("PRN_BENCH_TARGETS", "device"),
                    help="GT-only loss preparation: HIP kernels one batch ahead on the side stream (planerecnet_amd/targets.py, triplets drawn by the "
                         "device sampler) or the round-2 host worker processes (losses.TargetPrefetcher, numpy stream)")
    ap.add_argument("--hw-queues", type=int, default=None, help="GPU_MAX_HW_QUEUES for this run (default 3; applied before the HIP runtime starts)")
    ap.add_argument("--autotune", action="store_true", help="time the default launch-size threshold of the bf16-split GEMM kernel against the alternative on "
                    "this board before the warm-up (ops.autotune_split_policy).  Off by default: the board's clock answers over seconds, blocks of ten "
                    "steps under-estimate the penalty of the broad setting (50.4 vs 50.2 ms in-process where separate runs give 51.4 vs 50.1)")
    ap.add_argument("--no-exchange-probe", action="store_true", help="skip the extra steps that time the gradient-exchange path on a one-rank group")
    ap.add_argument("--graph", action="store_true", help="replay the network's forward / backward as two hipGraphs (measured SLOWER "
                    "than eager launches on ROCm 7.2 in round 1: 144.7 vs 134.3 ms/step -- ~2000 kernel nodes per replay.  Since the per-step operand refresh of the 16-bit-pipe launches "
                    "(ops.split_refresh_all) uploads its item table from pageable memory whenever the set of derived buffers changes -- which a capture's private pool makes happen -- "
                    "the capture is refused by the runtime and the run falls back to eager launches with a message; `hip_graph` in the line says which ran)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    args.config = args.config or wl[0]
    args.batch = args.batch or wl[1]
    args.height = args.height or wl[2]
    args.width = args.width or wl[3]
    train = wl[4]

    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher -- one process per GPU under torch.distributed.run
        # (the reference's single-process nn.DataParallel, train.py:153-213,266, is replaced by N replicas + RCCL)
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if os.environ.get("PRN_BENCH_ONE_DEVICE") or os.environ.get("PRN_ONE_DEVICE"):             # validation aid: N ranks time-share GPU 0 (with PRN_DIST_BACKEND=gloo;
        local = 0                                          # RCCL refuses two ranks on one device) -- exercises the N > 1 code path
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("PRN_FORCE_EXCHANGE"):  # (forced: a one-rank group, so that the bucket / RCCL path runs on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("PRN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from planerecnet_amd import ops, profiling, timer
    from planerecnet_amd.config import cfg, set_cfg
    from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher
    from planerecnet_amd.parallel import GradAllReduce, all_reduce_mean_scalars
    from planerecnet_amd.planerecnet import PlaneRecNet

    torch.set_num_threads(int(os.environ.get("PRN_HOST_THREADS", "4")))   # host-side tensor ops are small: a wide OpenMP team only adds fork/join latency
    timer.disable_all()                                # like the reference's train.py:233 (enabled timers synchronise per stage)
    set_cfg(args.config)
    if args.workload == "c5":
        cfg.replace({"max_size": 960})                 # BASELINE config 5: 4:3 frames -> 736 x 960 after pad_even_divided
    torch.manual_seed(0)                               # identical replicas on every rank
    net = PlaneRecNet(cfg)
    net.init_head_weights()
    images, inst, depths = synth_batch(args.batch, args.height, args.width, seed=1000 + rank, device=dev)
    np.random.seed(rank)
    ph, graphed, prefetch, losses = {}, False, None, None

    def set_dcn_offsets(px):
        """Non-zero offset / modulator conv weights (~px pixels r.m.s. offset; same rule as the parity tests' weights)."""
        from planerecnet_amd.dcn import DeformableConv2d
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, DeformableConv2d):
                    for c in (m.offset_conv, m.modulator_conv):
                        c.weight.copy_((torch.randn(c.weight.shape, generator=g) * px / (c.weight.shape[1] * 9) ** 0.5).to(c.weight.device))
                        c.bias.copy_((torch.randn(c.bias.shape, generator=g) * 0.1).to(c.bias.device))

    if train:
        net = net.to(dev).train()
        crit = PlaneRecNetLoss().to(dev)
        if os.environ.get("PRN_TORCH_ADAM"):                     # A/B: torch's multi-tensor Adam (14 launches) instead of the one-launch kernel
            opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
        else:
            from planerecnet_amd.optim import FusedAdam
            opt = FusedAdam(net.parameters(), lr=1e-4)
        exchange = GradAllReduce(list(net.parameters()), force=bool(os.environ.get("PRN_FORCE_EXCHANGE")))    # force: run the bucket / RCCL path with one rank too (overhead probe)
        if hasattr(opt, "exchange"):
            opt.exchange = exchange

        # --graph: the network's forward and backward are static, so each can be captured as ONE hipGraph
        # (torch.cuda.make_graphed_callables: stream capture of every HIP launch our C ABI issues on torch's current stream).
        # The loss (data-dependent shapes) and Adam stay eager.  Off by default: see the flag's help text.
        run_net = net
        if args.graph:
            try:
                run_net = torch.cuda.make_graphed_callables(net, (images,))
                graphed = True
            except Exception as e:                                   # noqa: BLE001
                print("bench.py: hipGraph capture failed (%s: %s); running eagerly" % (type(e).__name__, e), file=sys.stderr)
                run_net = net

        ops.set_wgrad_async(not args.sync_wgrad)                 # weight gradients on a side stream, joined after backward (ops.py)
        hw = (args.height, args.width)
        if args.targets == "device":
            from planerecnet_amd.targets import DeviceTargetBuilder
            prefetch = DeviceTargetBuilder(crit, seed=rank)
        else:
            prefetch = TargetPrefetcher(crit)
        prefetch.submit(inst, hw)                                # two batches in flight: the workers never wait for the trainer
        prefetch.submit(inst, hw)

        # Software-pipelined by one step: the targets of step k+1 are fetched (worker result, pinned staging, uploads on the side
        # stream) at the END of step k, so a step starts with the forward pass.  Every step still does exactly one get and one
        # submit; only the bubble after a device synchronisation shrinks (the GPU no longer idles through the 8 ms hand-over).
        pipe = {"targets": prefetch.get(depths, dev, overlap=True)}
        prefetch.submit(inst, hw)

        def step():
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            targets = pipe["targets"]
            if os.environ.get("PRN_BENCH_GAP"):                  # GPU time between the end of a step's Adam and the first launch of the next forward
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                ph.setdefault("_gap_b", []).append(ev)
            out = run_net(images)
            t3 = time.perf_counter()
            losses = crit(net, *out, inst, depths, targets=targets)
            loss = sum(losses.values()).sum()
            t4 = time.perf_counter()
            loss.backward()
            if os.environ.get("PRN_BENCH_GAP"):                  # how long does the compute stream wait for the weight-gradient stream?
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                ph.setdefault("_join_b", []).append(ev)
            ops.wgrad_join()
            if os.environ.get("PRN_BENCH_GAP"):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                ph.setdefault("_join_a", []).append(ev)
            exchange.finish()
            t5 = time.perf_counter()
            opt.step()
            if os.environ.get("PRN_BENCH_GAP"):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                ph.setdefault("_gap_a", []).append(ev)
                ev = torch.cuda.Event(enable_timing=True)           # (back to back with the previous one: what two adjacent records measure)
                ev.record()
                ph.setdefault("_gap_a2", []).append(ev)
            t6 = time.perf_counter()
            if os.environ.get("PRN_BENCH_PHASES") and args.targets == "workers":    # how long does `get` wait for the workers' results?
                tw = time.perf_counter()
                prefetch.queue[0][0].result()
                if prefetch.queue[0][1] is not None:
                    prefetch.queue[0][1].result()
                ph["get_wait"] = ph.get("get_wait", 0.0) + time.perf_counter() - tw
            if os.environ.get("PRN_BENCH_FIXED_TARGETS"):                       # (diagnostic only: the same targets every step, nothing fetched)
                t1 = t2 = time.perf_counter()
            else:
                pipe["targets"] = prefetch.get(depths, dev, overlap=True)      # GT-only targets of the NEXT step (prepared by the worker processes)
                t1 = time.perf_counter()
                prefetch.submit(inst, hw)                           # targets two steps further ahead: recomputed every step
                t2 = time.perf_counter()
            t1, t2 = t0 + (t1 - t6), t0 + (t2 - t6)             # (phase report: get / submit durations, fwd measured from t0)
            ph.setdefault("get_ms_per_step", []).append(round((t1 - t0) * 1e3, 1))
            ph.setdefault("host_ms_per_step", []).append(round((t6 - t0) * 1e3, 1))
            if os.environ.get("PRN_BENCH_PHASES"):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                ph.setdefault("_events", []).append(ev)
            for k, v in (("get", t1 - t0), ("submit", t2 - t1), ("fwd", t3 - t0), ("loss", t4 - t3), ("bwd", t5 - t4), ("adam", t6 - t5)):
                ph[k] = ph.get(k, 0.0) + v
            # (values only: a loss tensor returned with its graph would keep the whole step's autograd nodes -- and the buffers
            # the operators attach to them -- alive until the NEXT step has finished: +5 ms/step, 65.5 vs 60.5 ms)
            return {k: v.detach() for k, v in losses.items()}
    else:
        net = net.to(dev).eval()

        def step():                                             # one batch: eval forward (BatchNorm folded) + on-device post-process -> list[dict]
            with torch.no_grad():
                return net(images)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    BLOCK = 5
    blocks_log = {}

    def timed(n_warm, n_steps, tag=None):
        """n_warm untimed steps, then EXACTLY n_steps timed ones between two fences (the contract).  tag: also stamp the stream every BLOCK steps (event records,
        no synchronisation) -> blocks_log[tag] = ms per step of each block, read after the closing fence."""
        out = None
        for _ in range(n_warm):
            out = step()
        fence()
        marks = []
        t0 = time.perf_counter()
        for i in range(n_steps):
            if tag is not None and i % BLOCK == 0:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((i, ev))
            out = step()
        if tag is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((n_steps, ev))
        fence()
        el = time.perf_counter() - t0
        if tag is not None and len(marks) > 1:
            blocks_log[tag] = [round(a[1].elapsed_time(b[1]) / (b[0] - a[0]), 3) for a, b in zip(marks[:-1], marks[1:])]
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    def condition(max_steps=40, tol=0.01):
        """Untimed conditioning BEFORE the contract's warm-up: blocks of BLOCK steps until two consecutive blocks agree within `tol` (at most max_steps).
        A fresh lease pays its first-use costs here -- code-object loads of every kernel, the caching allocator's growth to the step's working set,
        the per-shape plans / descriptors / block states of the host layer, the board's clock ramp -- instead of inside the first timed steps
        (round 5: the driver's 5 + 20 steps read 45.7 ms where later legs of the same process read 42.2-43.4)."""
        ms, used = [], 0
        while used < max_steps:
            fence()
            t0 = time.perf_counter()
            for _ in range(BLOCK):
                step()
            fence()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            ms.append(round(1e3 * el / BLOCK, 3))
            used += BLOCK
            if len(ms) >= 2 and abs(ms[-1] - ms[-2]) <= tol * ms[-2]:
                break
        return ms

    # which plain GEMMs take the bf16-split kernel is a trade against this board's clock management: time the alternative (untimed steps,
    # before the warm-up proper; every rank decides on the slowest rank's times)
    tune = None
    if args.autotune and not graphed:
        def rmax(v):
            if world == 1:
                return v
            t = torch.tensor(v, device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.tolist()
        for _ in range(3):
            step()
        tune = ops.autotune_split_policy(step, "train" if train else "eval", reduce_max=rmax)
    conditioning = condition() if not os.environ.get("PRN_BENCH_NO_CONDITIONING") else []
    n_host0 = len(ph.get("host_ms_per_step", []))
    cpu0 = time.process_time()
    elapsed, last = timed(args.warmup, args.steps, tag="headline")
    # host side of the timed steps on this rank: wall time the trainer thread needs to enqueue a step, and CPU time of the whole
    # process (trainer + autograd + receiver threads; the target workers are separate processes) -- at N ranks per host these add up
    host = None
    if train:
        hs = ph.get("host_ms_per_step", [])[n_host0 + args.warmup:]
        cpu_free = 1e3 * (time.process_time() - cpu0) / max(args.steps + args.warmup, 1)
        # What the trainer needs to ENQUEUE a step: each probe step is issued while the stream is parked behind a spin kernel, so no launch waits for queue
        # space and no event wait absorbs slack -- in the free-running, GPU-bound loop above the host is throttled by the device (a full hardware queue blocks
        # inside the launches, the target hand-over waits for an event), and the wall time of a step's host phases then measures the GPU, not the host
        # (tools/wait_probe.py: 16-20 ms of host phases against 11 ms for the same work with the GPU parked).  Same steps, same process, after the timed leg.
        park, park_cpu = [], []
        for _ in range(0 if (graphed or os.environ.get("PRN_BENCH_NO_ENQUEUE_PROBE")) else 5):
            fence()
            torch.cuda._sleep(int(0.12 * 2.4e9))
            c0, t0 = time.process_time(), time.perf_counter()
            step()
            park.append(1e3 * (time.perf_counter() - t0))
            park_cpu.append(1e3 * (time.process_time() - c0))
        fence()
        med = lambda v: sorted(v)[len(v) // 2] if v else None      # noqa: E731
        host = {"enqueue_ms_per_step": med(park) if park else ((sum(hs) / len(hs)) if hs else None),
                "enqueue_probe": {"method": "whole step (zero_grad, forward, loss, backward, join, exchange, Adam, next targets) enqueued against a stream parked behind a "
                                            "spin kernel; median of the probes", "ms": [round(v, 2) for v in park], "process_cpu_ms": [round(v, 2) for v in park_cpu]},
                "free_running_trainer_ms_per_step": (sum(hs) / len(hs)) if hs else None,
                "process_cpu_ms_per_step": cpu_free,
                "process_cpu_note": "all threads of the process over the timed leg, free-running: trainer + autograd thread + the HIP / ROCr runtime's own threads "
                                    "(one of which polls completion signals at ~100 % of a core while the GPU is busy: tools/thread_probe.py)",
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "targets": args.targets,
                "target_prep_host_cpu_ms_per_step": (prefetch.host_cpu_ms / max(getattr(prefetch, "calls", 1), 1)) if hasattr(prefetch, "host_cpu_ms") else None,
                "target_prep_host_wall_ms_per_step": (prefetch.host_ms / max(getattr(prefetch, "calls", 1), 1)) if hasattr(prefetch, "host_ms") else None}
        if world > 1:                                            # slowest rank
            t = torch.tensor([host["enqueue_ms_per_step"] or 0.0, host["process_cpu_ms_per_step"]], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            host["enqueue_ms_per_step"], host["process_cpu_ms_per_step"] = float(t[0]), float(t[1])
    # cost of the gradient-exchange path itself (hooks, bucket pack, one-rank RCCL all-reduce, `.grad` re-pointing) before any wire
    # time: the same step with the exchange forced on a one-rank group (N = 1 only; with N > 1 it is always on)
    exch = None
    if train and world == 1 and rank == 0 and not graphed and not args.no_exchange_probe and not exchange.active:
        try:
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29534")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            plain = exchange
            exchange = GradAllReduce(list(net.parameters()), force=True)
            if hasattr(opt, "exchange"):
                opt.exchange = exchange
            n2 = max(args.steps // 2, 5)
            el2, _ = timed(3, n2)
            exchange.remove()
            exchange = plain
            if hasattr(opt, "exchange"):
                opt.exchange = plain
            el3, _ = timed(3, n2)                                # the plain step again, ADJACENT to the exchange leg: the difference below is between neighbours,
            exch = {"ms_per_step_with_exchange": 1e3 * el2 / n2,   # not against the headline leg minutes earlier (round 5 read -2.2 ms that way)
                    "ms_per_step_plain_adjacent": 1e3 * el3 / n2, "exchange_overhead_ms": max(1e3 * (el2 - el3) / n2, 0.0),
                    "exchange_overhead_raw_ms": 1e3 * (el2 - el3) / n2, "steps": n2,
                    "buckets": None, "note": "one-rank RCCL group: hooks + pack + all-reduce launch + unpack, no wire time"}
        except Exception as e:                                   # noqa: BLE001
            exch = {"error": "%s: %s" % (type(e).__name__, e)}
    if os.environ.get("PRN_SPLIT_STATS"):
        print("split-GEMM weight images: %s, %d cached operands" % (ops.SPLIT_STATS, len(ops._SPLIT_IMG)), file=sys.stderr)
    if os.environ.get("PRN_EXCHANGE_PROF"):
        from planerecnet_amd import parallel as _par
        n_ = max(_par._PROF.get("steps", 1), 1)
        print({k: (round(v / n_ * 1e3, 2) if isinstance(v, float) else v / n_) for k, v in _par._PROF.items()}, file=sys.stderr)
    if os.environ.get("PRN_BENCH_GAP") and train:
        ga, gb = ph.pop("_gap_a"), ph.pop("_gap_b")
        gaps = [a.elapsed_time(b) for a, b in zip(ga[:-1], gb[1:])]
        jb, ja = ph.pop("_join_b"), ph.pop("_join_a")
        print("GPU time the compute stream waits at wgrad_join (weight-gradient stream still busy): last ten %s" % [round(b.elapsed_time(a), 2) for b, a in zip(jb[-10:], ja[-10:])],
              file=sys.stderr)
        ga2 = ph.pop("_gap_a2")
        print("  of which between two adjacent event records: last ten %s" % [round(a.elapsed_time(b), 2) for a, b in zip(ga[-10:], ga2[-10:])], file=sys.stderr)
        print("GPU time from the end of Adam to the start of the next forward: mean %.2f ms, last ten %s" % (sum(gaps) / len(gaps), [round(g, 2) for g in gaps[-10:]]),
              file=sys.stderr)
    if os.environ.get("PRN_BENCH_PHASES") and train:
        evs = ph.pop("_events")
        ph["gpu_ms_per_step"] = [round(a.elapsed_time(b), 1) for a, b in zip(evs[:-1], evs[1:])]
        print({k: (v if isinstance(v, list) else round(v / (args.steps + args.warmup) * 1e3, 2)) for k, v in ph.items()}, file=sys.stderr)
    if train:
        losses = last
        loss_means = all_reduce_mean_scalars([losses[k].detach().sum() for k in sorted(losses)], dev).tolist()
        finite = all(np.isfinite(v) for v in loss_means)
        detections = None
    else:
        loss_means, finite = None, all(bool(torch.isfinite(r["pred_depth"]).all()) for r in last)
        detections = sum(0 if r["pred_scores"] is None else len(r["pred_scores"]) for r in last)

    roof, kernels = None, None
    # (EVERY rank runs this leg: its steps contain the gradient exchange and `timed` ends in a barrier and a reduction -- rank 0 alone would issue
    # collectives nobody answers.  Only rank 0's brackets are reported.)
    if not args.no_roofline:
        if graphed:
            raise SystemExit("bench.py: the roofline leg needs eagerly issued launches; combine --graph with --no-roofline")
        branch_streams, ops.BRANCH_STREAMS = ops.BRANCH_STREAMS, False      # one stream: an event pair then times one launch, not its neighbours
        was_async = ops.WGRAD_ASYNC
        if train:
            ops.set_wgrad_async(False)                               # every kernel alone and planned as a stand-alone launch (ops.WGRAD_WGS_ASYNC)
        profiling.enable()
        # Park the stream behind a ~150 ms spin kernel while the host enqueues the bracketed step: with two event records per
        # launch the host is slower than the GPU, and every event pair would otherwise also time the idle wait for the
        # kernel's submission (several us per launch, i.e. 5-10 % of the small kernels' durations).
        torch.cuda._sleep(int(0.15 * 2.4e9))
        step()
        torch.cuda.synchronize()
        fams = profiling.summary()
        if os.environ.get("PRN_BENCH_SHAPES") and rank == 0:    # per-shape table of the MFMA launches of the bracketed step
            with open(os.environ["PRN_BENCH_SHAPES"], "w") as fh:
                fh.write("%-22s %-44s %6s %9s %8s %7s\n" % ("family", "(kind, C, H, W, M, K, stride, mode, dil, B)", "calls", "ms/step", "us/call", "TF/s"))
                for fam, tag, n, ms, work in profiling.by_shape():
                    if tag is not None:
                        fh.write("%-22s %-44s %6d %9.3f %8.1f %7.1f\n" % (fam, str(tag), n, ms, ms / n * 1e3, work / (ms * 1e-3) / 1e12))
        profiling.disable()
        if train:
            ops.set_wgrad_async(was_async)
        ops.BRANCH_STREAMS = branch_streams
        kernels = fams
        dom = max((f for f in fams if f["bound"] == "mfma"), key=lambda f: f["time_ms"])
        pmc = pmc_traffic(dom["kernel"])
        traffic = None
        if pmc is not None or dom.get("bytes"):
            alg = dom["bytes"] / dom["launches"] if dom.get("bytes") else None
            traffic = {"hbm_bytes_per_launch": None if pmc is None else pmc["bytes_per_launch"], "source": None if pmc is None else pmc["source"],
                       "algorithmic_bytes_per_launch": alg,
                       "ratio": (pmc["bytes_per_launch"] / alg) if (pmc is not None and alg) else None}
        # the same comparison for every MFMA family that logs algorithmic bytes (weight gradients, the fused DCNv2 launches)
        by_family = {}
        for f in fams:
            if f["bound"] == "mfma" and f.get("bytes"):
                pm = pmc_traffic(f["kernel"])
                alg_f = f["bytes"] / f["launches"]
                by_family[f["kernel"]] = {"algorithmic_bytes_per_launch": alg_f, "hbm_bytes_per_launch": None if pm is None else pm["bytes_per_launch"],
                                          "ratio": None if pm is None else pm["bytes_per_launch"] / alg_f, "source": None if pm is None else pm["source"]}
        if traffic is not None:
            traffic["by_family"] = by_family
            traffic["note"] = ("counter bytes come from a step run with --sync-wgrad (one kernel at a time): weight gradients are then launched one layer at a "
                               "time with ~48 pixel splits each, not eight layers per launch as in the timed step -- their ratio is an upper bound (DESIGN.md 10.5b)")
        # HBM-bound families are credited with the bytes their kernels EXECUTE (per variant), so none can exceed what the memory
        # system delivers; a figure above the measured copy rate means the accounting of that family is wrong
        over = [f["kernel"] for f in fams if f["bound"] == "hbm" and f["achieved"] > 6300.0 and f["time_ms"] > 0.02]
        # ALGORITHMIC FLOPs (2*M*N*K of the fp32 GEMMs a launch evaluates) over the launch time, against the dense matrix peak of the dtype the
        # path computes in (fp32: 157.3 TFLOP/s).  For the fp32 MFMA kernels that is also what they execute; the split kernel executes three
        # (fp16 pieces) or six (bf16 pieces) products per multiply-add on the 16-bit pipe -- its pipe-side view is in `split_gemm` below.
        dom_alg = dom["achieved"] / (ops.split_products() if dom["kernel"] in PIPE16 else 1.0)
        roof = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom_alg, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": dom_alg / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "launches": dom["launches"],
                "avg_launch_us": 1e3 * dom["time_ms"] / dom["launches"],
                "flops_per_launch": dom["work"] / dom["launches"] / (ops.split_products() if dom["kernel"] in PIPE16 else 1.0),
                "hbm_families_above_copy_rate": over}
        if dom["kernel"] in PIPE16:
            # The dominant family issues 16-bit piece products.  Pricing its ALGORITHMIC fp32 FLOPs against the fp32 MFMA peak (0.9+) would flatter it:
            # it does not run on that pipe.  The roofline that binds it is the larger of the two time bounds of an average launch -- executed piece
            # products over the 16-bit pipe's dense peak, algorithmic bytes over the HBM peak; for these short-K layers (5 GFLOP against ~90 MB)
            # that is the MEMORY side.  `frac` is against that bound; the other views stay next to it.
            t_mfma = dom["work"] / (PEAK_BF16_MFMA_TFLOPS * 1e12)
            t_hbm = dom["bytes"] / (PEAK_HBM_GBS * 1e9) if dom.get("bytes") else 0.0
            roof["fp32_equivalent"] = {"achieved": dom_alg, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom_alg / PEAK_FP32_MFMA_TFLOPS,
                                       "note": "algorithmic fp32 FLOPs over the fp32 MFMA peak: what the same launches would need on the fp32 pipe"}
            roof["executed_on"] = "fp16 matrix pipe (%g piece products per multiply-add)" % ops.split_products()
            roof["executed_achieved"] = dom["achieved"]
            roof["executed_peak"] = PEAK_BF16_MFMA_TFLOPS
            roof["executed_frac"] = dom["achieved"] / PEAK_BF16_MFMA_TFLOPS
            roof["time_bounds_us_per_launch"] = {"mfma_16bit_pipe": 1e6 * t_mfma / dom["launches"], "hbm": 1e6 * t_hbm / dom["launches"]}
            if t_hbm > t_mfma:
                gbs = dom["bytes"] / (dom["time_ms"] * 1e-3) / 1e9
                roof.update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
            else:
                roof.update({"achieved": dom["achieved"], "peak": PEAK_BF16_MFMA_TFLOPS, "frac": dom["achieved"] / PEAK_BF16_MFMA_TFLOPS})
        # `achieved` above credits every launch with the FLOPs it EXECUTES (an MFMA utilisation) -- that is the headline
        # figure.  Footnote: the Winograd launches execute a quarter of the multiply-adds of the convolution they evaluate, so the
        # same step is also summarised as FLOPs of the REFERENCE convolutions over the time of every kernel of the convolution
        # family, transforms and reductions included.
        # The plain-GEMM launches that run on the bf16 matrix pipe by exact operand splitting (csrc/prn_gemm_split.hip): `achieved` /
        # `frac` price the six bf16 piece products each fp32 multiply-add costs against the bf16 peak; `fp32_equivalent` is the fp32
        # GEMM rate the launches deliver (2*M*N*K over their time), next to the fp32 MFMA peak it would otherwise be bounded by.
        for f in [f for f in fams if f["kernel"] in PIPE16]:
            eq = f["work"] / ops.split_products() / (f["time_ms"] * 1e-3) / 1e12     # 2*M*N*K of the GEMMs the launches evaluate
            roof["split_gemm" if f["kernel"] == "split_gemm_kernel" else "wgrad16"] = {"launches": f["launches"], "time_ms": f["time_ms"], "achieved": f["achieved"], "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": f["achieved"] / PEAK_BF16_MFMA_TFLOPS, "fp32_equivalent": eq, "fp32_equivalent_over_fp32_mfma_peak": eq / PEAK_FP32_MFMA_TFLOPS,
                                  "pieces": ("2 x fp16 per fp32 operand (scaled by exact powers of two per weight row / activation column), 3 of 4 products, fp32 accumulate"
                                             if ops.split_products() <= 4.0 else "3 x bf16 per fp32 operand (exact), 6 of 9 products, fp32 accumulate")}
        # The forward / input-gradient GEMM launches as ONE family, whichever pipe a launch took: 2*M*N*K of the GEMMs they evaluate over
        # their time, against the fp32 MFMA peak -- comparable with the conv_igemm figure of the rounds before the split kernel existed (the
        # launches that moved to it were conv_igemm's most efficient ones, so that family's own `frac` falls when they leave).
        gm = [f for f in fams if f["kernel"] in ("conv_igemm_kernel", "split_gemm_kernel")]
        if len(gm) == 2:
            fl = sum(f["work"] / (ops.split_products() if f["kernel"] in PIPE16 else 1.0) for f in gm)
            ms = sum(f["time_ms"] for f in gm)
            roof["gemm_launches_fp32_equivalent"] = {"kernels": [f["kernel"] for f in gm], "launches": sum(f["launches"] for f in gm), "time_ms": ms,
                                                     "achieved": fl / (ms * 1e-3) / 1e12, "unit": "TFLOP/s", "peak": PEAK_FP32_MFMA_TFLOPS,
                                                     "frac": fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        conv = [f for f in fams if f["kernel"] in ("conv_igemm_kernel", "split_gemm_kernel", "reduce_epilogue_kernel", "winograd_input_kernel", "winograd_output_kernel",
                                                    "conv3x3_winograd_ragged", "conv_wgrad_kernel", "wgrad16_kernel", "reduce_splits_kernel", "winograd_wgrad_transforms",
                                                    "winograd_dw_kernel", "conv3x3_winograd_wgrad_ragged", "dcnv2_fwd_kernel", "dcnv2_wgrad_kernel")]
        ref_flops = sum(f["ref_work"] for f in conv if f["bound"] == "mfma")
        conv_ms = sum(f["time_ms"] for f in conv)
        # (the bracketed leg brackets every launch on its own, so it runs with the block entry points off (blocks.usable is false under the
        # profiler): every operator writes its own result there, ~160 launches more than the timed step issues; the MFMA families the roofline is about are the same launches)
        roof["bracketed_leg"] = "every operator writes its own result (the timed step's producer -> BatchNorm hand-overs are off while launches are bracketed one by one)"
        roof["footnote_reference_operator_view"] = {"reference_flops_per_step": ref_flops, "time_ms": conv_ms,
                                                    "achieved": ref_flops / (conv_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                                                    "frac": ref_flops / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        # the DCNv2 operator on its own (north_star names it): fused forward + weight-gradient GEMMs against the MFMA peak
        dcn = [f for f in fams if f["kernel"] in ("dcnv2_fwd_kernel", "dcnv2_wgrad_kernel")]
        if dcn:
            roof["dcnv2"] = {f["kernel"]: {"achieved": f["achieved"], "unit": f["unit"], "frac": f["achieved"] / PEAK_FP32_MFMA_TFLOPS,
                                           "launches": f["launches"], "avg_launch_us": 1e3 * f["time_ms"] / f["launches"]} for f in dcn}

        # north star: ">= 60 % of the relevant roofline on the DCNv2 + conv kernels" -- one line per kernel group, all as ALGORITHMIC fp32 FLOPs
        # over launch time against the fp32 MFMA peak (the dominant-kernel object above is one of these rows, whichever is largest by time)
        def _alg(f):
            return f["achieved"] / (ops.split_products() if f["kernel"] in PIPE16 else 1.0)
        groups = {"forward_and_input_gradient_gemms": ("conv_igemm_kernel", "split_gemm_kernel"), "weight_gradient_gemms": ("conv_wgrad_kernel", "wgrad16_kernel"),
                  "dcnv2_forward": ("dcnv2_fwd_kernel",), "dcnv2_weight_gradient": ("dcnv2_wgrad_kernel",)}
        view = {}
        for name, ks in groups.items():
            fs = [f for f in fams if f["kernel"] in ks and f["time_ms"] > 0]
            if fs:
                ms = sum(f["time_ms"] for f in fs)
                tf = sum(_alg(f) * f["time_ms"] for f in fs) / ms
                view[name] = {"time_ms": ms, "launches": sum(f["launches"] for f in fs), "achieved": tf, "frac": tf / PEAK_FP32_MFMA_TFLOPS}
        roof["conv_kernel_groups"] = view

    # the same step with every contraction on the fp32 MFMA kernels (same weights, same process, after the headline timing)
    fp32_run = None
    if not graphed and ops.split_mode() != 0 and not os.environ.get("PRN_BENCH_NO_FP32_RUN"):
        old_pol = ops.set_split_gemm(mode=0)
        n2 = max(args.steps // 2, 5)
        el2, _ = timed(3, n2)
        ops.set_split_gemm(**old_pol)
        fp32_run = {"steps": n2, "ms_per_step": 1e3 * el2 / n2, "value": args.batch * world * n2 / el2,
                    "arithmetic": "every contraction on v_mfma_f32_32x32x2_f32 (fp32 operands, fp32 accumulate); nothing on the 16-bit pipe"}
        timed(2, 0)                                              # (back on the default plan before the legs below)

    # inference workloads: the timed post-process at the init state sees (almost) no candidates.  Shift the category bias until every image
    # keeps >= 5 detections and time the same batch again (the product's own forward picks the shift: no oracle on this path).
    cond_run = None
    if not train and not os.environ.get("PRN_BENCH_NO_CONDITIONED_RUN"):       # (every rank: `timed` synchronises the ranks; the shift is agreed by a MIN over ranks)
        bias = net.inst_head.cate_pred.bias
        b0 = bias.detach().clone()
        for shift in (0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0):
            with torch.no_grad():
                bias.copy_(b0 + shift)
            res = step()                                         # (in-place copy_: version counters move, cached derived weights follow)
            counts = [0 if r["pred_scores"] is None else len(r["pred_scores"]) for r in res]
            least = torch.tensor([min(counts)], device=dev, dtype=torch.int64)
            if world > 1:                                        # every rank draws its own batch (seed 1000 + rank): the decision to stop at this
                dist.all_reduce(least, op=dist.ReduceOp.MIN)     # shift -- and to enter `timed`, which ends in collectives -- is taken by ALL ranks together
            if int(least) >= 5:
                n2 = max(args.steps // 2, 5)
                el2, res = timed(2, n2)
                cond_run = {"cate_bias_shift": shift, "detections_per_image": [0 if r["pred_scores"] is None else len(r["pred_scores"]) for r in res],
                            "steps": n2, "ms_per_step": 1e3 * el2 / n2, "value": args.batch * world * n2 / el2}
                break
        with torch.no_grad():
            bias.copy_(b0)

    # the same step with non-trivial deformable offsets (after the headline timing; the weights change)
    dcn_run = None
    if train and args.dcn_offsets > 0 and not graphed:
        set_dcn_offsets(args.dcn_offsets)
        n2 = max(args.steps // 2, 3)
        el2, l2 = timed(2, n2)
        dcn_run = {"offset_px_rms": args.dcn_offsets, "steps": n2, "ms_per_step": 1e3 * el2 / n2, "value": args.batch * world * n2 / el2,
                   "losses_finite": all(bool(torch.isfinite(v).all()) for v in l2.values())}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.config, args.height, args.width, train, batch=args.batch)

    # The JSON line must be the LAST thing on the job's stdout: RCCL writes its version banner through C stdio, which sits in the
    # process's buffer until exit when stdout is a pipe -- i.e. it would follow the line.  Every rank flushes C stdio now, rank 0
    # prints after a barrier, and then every rank points fd 1 at /dev/null for the teardown.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if world > 1:
        dist.barrier()
    if rank == 0:
        gb = args.batch * world
        line = {"metric": METRIC, "value": gb * args.steps / elapsed, "unit": "img/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None,
                "dtype": ("f32" if ops.split_mode() == 0 else
                          "f32 (tensors and accumulation fp32; plain GEMMs and the tap-walked 4x4/s2, 1x1/s2 and sub-pixel convolutions >= %d tiles: %s split MFMA, fp32 accumulate; fp32-only rate in fp32_only_run)"
                          % (ops._POLICY["min_tiles"], "2xfp16-piece" if ops.split_products() <= 4.0 else "3xbf16-piece")),
                "data": "synthetic",
                "arithmetic": "fp32 tensors, fp32 accumulation, fp32 MFMA; plain GEMMs (and, since round 5, the 4x4 / stride-2, 1x1 / stride-2 and sub-pixel-phase convolutions, walked tap by tap) of >= %d output tiles run on the 16-bit matrix pipe with every fp32 operand cut "
                              "into %s (error vs fp64 at the fp32 MFMA's level; DESIGN.md 9.1b / 9.1c; PRN_SPLIT_GEMM=0 turns it off)"
                              % (ops._POLICY["min_tiles"], "two fp16 pieces after an exact power-of-two scaling" if ops.split_products() <= 4.0 else "three exact bf16 pieces"),
                "config": {"workload": "%s: %s %s, per-GPU batch %d, %dx%d synthetic %s, random-init weights"
                           % (args.workload, args.config, wl[5], args.batch, args.height, args.width, "RGB+depth+planes" if train else "RGB"),
                           "global_batch": gb, "parallelism": ("dp%d" % world) if train else ("replicas%d" % world)},
                "hip_graph": graphed, "losses_finite": finite, "losses": None if loss_means is None else dict(zip(sorted(losses), loss_means)),
                "detections_last_batch": detections, "roofline": roof, "cpu_baseline": cpu, "fp32_only_run": fp32_run, "conditioned_run": cond_run,
                "dcn_offsets_run": dcn_run, "host": host,
                # steady state: untimed conditioning blocks (ms per step) before the contract's warm-up, and the timed leg resolved into blocks of five steps
                # (stream stamps, no synchronisation inside the leg): the headline is the whole leg; its blocks' median / min / max show whether it was steady
                "steady_state": {"conditioning_blocks_ms_per_step": conditioning, "timed_blocks_ms_per_step": blocks_log.get("headline"),
                                 "timed_blocks_median_min_max": (lambda b: [sorted(b)[len(b) // 2], min(b), max(b)] if b else None)(blocks_log.get("headline")),
                                 "block_steps": BLOCK},
                "exchange_probe": exch, "split_gemm_policy": tune,
                "device": {"name": torch.cuda.get_device_name(dev), "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", None))},   # boxes differ by +-2 %
                "kernels": kernels}
        print(json.dumps(line), flush=True)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    if prefetch is not None:
        prefetch.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
