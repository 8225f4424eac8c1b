#!/bin/bash
# rocprofv3 kernel stats of an inference workload (c1 | c2 | c5): launches and time per kernel family per batch
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-c1}
rm -rf /tmp/prof_inf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o t -- python $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /tmp/prof_inf.log 2>&1
tail -1 /tmp/prof_inf.log | cut -c1-200
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/prof_inf/*kernel_stats.csv')[0])))
steps = 25.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
n = sum(int(r["Calls"]) for r in rows) / steps
print("%.1f us of kernels / batch, %.0f launches / batch" % (tot, n))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%8.1f us/batch %6.1f calls/batch avg %7.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e3, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
