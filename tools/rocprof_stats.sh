#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py: (1) serialised (--sync-wgrad: every kernel alone on the GPU -- the run whose
# per-kernel averages match the event-bracketed roofline leg) and (2) the default overlapped run.  Summaries -> gpurun_out/prof/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
mkdir -p $R/gpurun_out/prof
for mode in serial default; do
  rm -rf /tmp/prof_$mode
  extra=""; [ $mode = serial ] && extra="--sync-wgrad"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o t -- env PRN_BENCH_NO_FP32_RUN=1 PRN_BENCH_NO_ENQUEUE_PROBE=1 python $R/bench.py --no-exchange-probe --steps 10 --warmup 1 --no-cpu-baseline --no-roofline --dcn-offsets 0 $extra > /tmp/prof_$mode.log 2>&1
  f=$(ls /tmp/prof_$mode/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/prof/${TAG}_kernel_stats_$mode.csv
  f=$(ls /tmp/prof_$mode/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f /tmp/prof_${mode}_trace.csv
  tail -2 /tmp/prof_$mode.log | cut -c1-300
done
python3 - <<PY
import csv
for mode in ("serial", "default"):
    rows = list(csv.DictReader(open("$R/gpurun_out/prof/${TAG}_kernel_stats_%s.csv" % mode)))
    # steps actually executed (warm-up, timed, the exchange probe and the roofline-free extras): the one-launch Adam runs once per step
    adam = [int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"]]
    steps = adam[0] if adam else 11
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
    n = sum(int(r["Calls"]) for r in rows) / steps
    ours = [r for r in rows if "anonymous namespace" in r["Name"] and "at::" not in r["Name"]]
    print("%s: %.2f ms of kernels / step, %.0f launches / step; libprn_hip: %.2f ms, %.0f launches" % (mode, tot, n, sum(float(r["TotalDurationNs"]) for r in ours) / steps / 1e6, sum(int(r["Calls"]) for r in ours) / steps))
    fam = {}
    for r in rows:
        nm = r["Name"]
        key = "ATen / rocprim / copies"
        for k in ("wgrad16_kernel", "split16_gemm_kernel", "split_gemm_kernel", "split16_", "split_prepare", "conv_igemm_kernel", "conv_wgrad_kernel", "dcnv2_fwd", "dcnv2_wgrad_kernel", "dcnv2_table", "dcnv2_patch_reduce", "dcn_", "winograd_", "bn_", "gn_relu", "reduce_splits", "reduce_epilogue", "resize_", "mask_loss", "maxpool", "channel_sum", "flip_transpose", "pad_fold", "replicate_fold", "space_to_depth", "up2_", "conv3x3_"):
            if k in nm and "at::" not in nm:
                key = k
                break
        a = fam.setdefault(key, [0, 0.0])
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("   %-28s %7.1f launches/step %8.3f ms/step  avg %7.1f us" % (k, c / steps, t / steps / 1e6, t / c / 1e3))
# Steady-state launch counts: the stats above divide the WHOLE run by the number of steps, so one-time work (the model's upload: ~850 buffer copies; Adam's state:
# ~1100 zero fills; weight initialisation) is spread over the steps.  From the kernel trace: the dispatches between two Adam launches five steps apart, late in the run.
import os
for mode in ("serial", "default"):
    f = "/tmp/prof_%s_trace.csv" % mode
    if not os.path.exists(f):
        continue
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    if len(adam) < 8:
        continue
    a, b = adam[-7], adam[-2]
    win = rows[a + 1:b + 1]
    mine = lambda r: "anonymous namespace" in r["Kernel_Name"] and "at::" not in r["Kernel_Name"]
    ours = [r for r in win if mine(r)]
    other = {}
    for r in win:
        if mine(r):
            continue
        nm = r["Kernel_Name"]
        key = nm.split("<")[0].split("(")[0][-60:] if "rocclr" in nm or "rocprim" in nm else nm[:100]
        other[key] = other.get(key, 0) + 1
    print("%s, steady state (5 steps between Adam launches): %.1f launches / step; libprn_hip %.1f, ATen / rocprim / runtime copies %.1f" % (mode, len(win) / 5, len(ours) / 5, (len(win) - len(ours)) / 5))
    if mode == "default":
        for k, c in sorted(other.items(), key=lambda kv: -kv[1])[:25]:
            print("      %6.1f / step  %s" % (c / 5, k))
PY
