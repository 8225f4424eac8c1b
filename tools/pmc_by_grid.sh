#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one kernel family per launch, grouped by grid size (one training step):  tools/pmc_by_grid.sh wgrad16_kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-wgrad16_kernel}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pg_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pg_$c -o t -- env PRN_BENCH_NO_FP32_RUN=1 python $R/bench.py --no-exchange-probe --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --sync-wgrad --dcn-offsets 0 > /tmp/pg_$c.log 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, collections, sys
K = sys.argv[1]
acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob('/tmp/pg_%s/*counter_collection.csv' % c)[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c and K in r['Kernel_Name']:
            acc[int(r['Grid_Size'])][c].append(float(r['Counter_Value']))
print("%s: per launch by grid size (2 steps profiled); fetch x2 (gfx950 correction), MB" % K)
for g in sorted(acc, key=lambda g: -sum(acc[g]["FETCH_SIZE"])):
    f, w = acc[g]["FETCH_SIZE"], acc[g]["WRITE_SIZE"]
    print("  grid %8d (%5d workgroups)  launches %3d   fetch %8.1f MB   write %8.1f MB" % (g, g // 256, len(f), 2 * sum(f) / len(f) / 1024, sum(w) / max(len(w), 1) / 1024))
PY
