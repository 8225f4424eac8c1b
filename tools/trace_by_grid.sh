#!/bin/bash
# Kernel durations of one family grouped by grid size (serial training step):  tools/trace_by_grid.sh "resize|fold|space_to_depth|maxpool"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=${1:-resize}
rm -rf /tmp/tg
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- env PRN_BENCH_NO_FP32_RUN=1 python $R/bench.py --no-exchange-probe --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --sync-wgrad --dcn-offsets 0 > /tmp/tg.log 2>&1
python3 - "$PAT" <<'PY'
import csv, glob, collections, re, sys
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob('/tmp/tg/*kernel_trace.csv')[0])):
    n = r['Kernel_Name']
    if pat.search(n):
        n = n.replace('(anonymous namespace)::', '').split('(')[0]
        acc[(n, int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
steps = 4.0
print("%-28s %12s %8s %9s %10s" % ("kernel", "threads", "calls/st", "avg us", "ms/step"))
for (n, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s %12d %8.1f %9.1f %10.3f" % (n[:28], g, len(v) / steps, sum(v) / len(v), sum(v) / steps / 1e3))
PY
