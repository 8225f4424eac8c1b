#!/bin/bash
# How much of a training step is the GPU idle?  rocprofv3 --kernel-trace of the default (overlapped) bench run; busy time =
# union of the kernel intervals over all streams, per step window (steps are delimited by the fused-Adam launches).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/trace_idle
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_idle -o t -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --dcn-offsets 0 > /tmp/trace_idle.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/trace_idle/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
adam = [s for s, e, n in rows if 'FusedAdam' in n or 'fused_adam' in n.lower()]
# one step = from the first Adam kernel of step k to the first Adam kernel of step k+1 (14 Adam launches per step)
marks = adam[::14]
print("steps seen:", len(marks) - 1)
res = []
for a, b in zip(marks[4:-1], marks[5:]):
    iv = [(s, e) for s, e, n in rows if s >= a and s < b]
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = []
    last = None
    for s, e in iv:
        if last is not None and s > last: gaps.append(s - last)
        last = e if last is None else max(last, e)
    big = sorted(gaps, reverse=True)[:5]
    res.append(((b - a) / 1e6, busy / 1e6, len(iv), [round(g / 1e3) for g in big]))
for r in res: print("step %.2f ms  busy %.2f ms  idle %.2f ms  kernels %d  largest gaps (us) %s" % (r[0], r[1], r[0] - r[1], r[2], r[3]))
PY
