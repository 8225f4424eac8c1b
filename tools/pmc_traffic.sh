#!/bin/bash
# HBM traffic per kernel family for one bench step: two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass).
# Prints per-family average bytes/launch plus the calibration ratios on kernels whose algorithmic bytes are known.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o t -- env PRN_BENCH_NO_FP32_RUN=1 PRN_BENCH_NO_ENQUEUE_PROBE=1 PRN_BENCH_NO_CONDITIONING=1 python $R/bench.py --no-exchange-probe --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --sync-wgrad --dcn-offsets 0 > /tmp/pmc_$c.log 2>&1
  ls /tmp/pmc_$c | head -5
done
python3 - <<'PY'
import csv, glob, collections, json, os
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob('/tmp/pmc_%s/*counter_collection.csv' % c)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != c: continue
        k = r['Kernel_Name']
        fam = None
        for key in ("dcnv2_fwd", "dcnv2_wgrad_kernel", "dcnv2_table", "dcnv2_patch_reduce", "mask_loss_partial", "mask_loss_bwd", "vnl_triplet_kernel", "vnl_scatter_kernel", "wgrad16_kernel", "split16_gemm_kernel", "split16_prepare_batched_kernel", "split16_rowmax_batched_kernel", "split_gemm_kernel", "split_prepare_kernel", "conv_igemm_kernel", "conv_wgrad_kernel", "reduce_epilogue_kernel", "reduce_splits_kernel", "bn_small_fwd_kernel", "bn_small_bwd_kernel", "bn_partial_kernel", "bn_apply_kernel", "bn_bwd_partial_kernel", "bn_bwd_apply_kernel", "dcn_dom_partial_kernel", "dcn_dx_gather_kernel", "dcn_csr_fill_kernel", "dcn_csr_count_kernel", "dcn_sample_kernel", "gn_relu_fwd", "gn_relu_bwd", "resize_fwd", "resize_bwd", "flip_transpose_batched", "space_to_depth2", "replicate_fold", "winograd_input_kernel", "winograd_output_kernel", "winograd_dy_kernel", "winograd_dw_kernel", "winograd_weights_kernel"):
            if key in k: fam = key; break
        if fam: acc[fam].append(float(r['Counter_Value']))
    out[c] = {k: {"launches": len(v), "sum_kb": sum(v), "avg_kb": sum(v)/len(v)} for k, v in acc.items()}
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
json.dump(out, open(root + "/gpurun_out/pmc_traffic/summary.json", "w"), indent=1)
# the file bench.py reads (copy to profiles/<round>_pmc_traffic.json): bytes per launch with the gfx950 fetch correction
STEPS = 2
fams = {}
for k in set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"]):
    f, w = out["FETCH_SIZE"].get(k), out["WRITE_SIZE"].get(k)
    if not f or not w: continue
    fams[k] = {"launches_per_step": f["launches"] / STEPS, "fetch_kb_avg": f["avg_kb"], "write_kb_avg": w["avg_kb"],
               "hbm_bytes_per_launch": (2.0 * f["avg_kb"] + w["avg_kb"]) * 1024.0}
note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) over `bench.py --steps 1 --warmup 1 --sync-wgrad` "
        "(= 2 steps), tools/pmc_traffic.sh. Units: KB as reported. gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per "
        "128-B request -> x2 (calibrated in round 1 on bn_apply / bn_partial, whose bytes are known: measured/known = 0.48); WRITE_SIZE as is. "
        "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024.")
json.dump({"_note": note, "steps_profiled": STEPS, "families": fams}, open(root + "/gpurun_out/pmc_traffic/pmc_traffic.json", "w"), indent=1)
for c, d in out.items():
    print(c)
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["sum_kb"]):
        print("  %-26s launches %5d  total %10.1f MB   avg %9.1f KB" % (k, v["launches"], v["sum_kb"]/1024, v["avg_kb"]))
PY
