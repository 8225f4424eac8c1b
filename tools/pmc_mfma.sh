#!/bin/bash
# MFMA-pipe utilisation per kernel family over bench steps: one PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).
# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): the gfx94x derived-metric formula; rocprofv3
# reports GRBM_GUI_ACTIVE summed over the 8 XCDs (1.46 M per 75 us launch = 8 x 180 k cycles), SQ counters summed over all SIMDs.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
rm -rf /tmp/pmc_mfma
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -o t -- env PRN_BENCH_NO_FP32_RUN=1 PRN_BENCH_NO_ENQUEUE_PROBE=1 PRN_BENCH_NO_CONDITIONING=1 python $R/bench.py --no-exchange-probe --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --sync-wgrad --dcn-offsets 0 > /tmp/pmc_mfma.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json, os
f = glob.glob('/tmp/pmc_mfma/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    fam = None
    for key in ("wgrad16_kernel", "split16_gemm_kernel", "split_gemm_kernel", "conv_igemm_kernel", "conv_wgrad_kernel", "dcnv2_fwd", "dcnv2_wgrad_kernel"):
        if key in k:
            fam = key
    if fam is None:
        continue
    acc[fam][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        cnt[fam] += 1
out = {}
for fam, d in acc.items():
    util = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8.0 * 256 * 4)
    out[fam] = {"launches": cnt[fam], "SQ_VALU_MFMA_BUSY_CYCLES": d['SQ_VALU_MFMA_BUSY_CYCLES'], "GRBM_GUI_ACTIVE": d['GRBM_GUI_ACTIVE'], "mfma_util": util}
    print(fam, cnt[fam], "launches, MfmaUtil %.1f %%" % (100 * util))
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
json.dump({"_note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 1 --warmup 1 --sync-wgrad` (2 steps); "
                    "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), summed over the launches of the family", "families": out},
          open(root + "/gpurun_out/pmc_mfma.json", "w"), indent=1)
PY
