#!/bin/bash
# Which kernels does the gradient exchange add to a step?  rocprofv3 kernel stats of the one-rank forced exchange (PRN_FORCE_EXCHANGE=1) against the plain step.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in plain exchange; do
  rm -rf /tmp/prof_$mode
  extra=""; [ $mode = exchange ] && extra="PRN_FORCE_EXCHANGE=1"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o t -- env PRN_BENCH_NO_FP32_RUN=1 PRN_BENCH_NO_ENQUEUE_PROBE=1 PRN_BENCH_NO_CONDITIONING=1 $extra python $R/bench.py --no-exchange-probe --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --dcn-offsets 0 > /tmp/prof_$mode.log 2>&1
  cp $(ls /tmp/prof_$mode/*kernel_stats.csv | head -1) $R/gpurun_out/exchange_$mode.csv
done
python3 - <<PY
import csv
def load(m):
    rows = list(csv.DictReader(open("$R/gpurun_out/exchange_%s.csv" % m)))
    steps = [int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"]][0]
    return {r["Name"]: (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6) for r in rows}, steps
a, sa = load("plain"); b, sb = load("exchange")
print("steps", sa, sb, " total kernel ms/step plain %.2f exchange %.2f" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
d = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], b.get(k, (0, 0))[0] - a.get(k, (0, 0))[0], k) for k in set(a) | set(b)), reverse=True)
for dt, dc, k in d[:14]:
    print("%+8.3f ms %+7.1f launches  %s" % (dt, dc, k[:120]))
PY
