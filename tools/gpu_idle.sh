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
adam = [s for s, e, n in rows if 'adam_kernel' in n]
# one step = from the Adam launch of step k to the Adam launch of step k+1 (prn_adam_step: one per step)
marks = adam
print("steps seen:", len(marks) - 1)
res = []
where = []
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
    last, last_name = None, None
    named = [(s, e, n) for s, e, n in rows if s >= a and s < b]
    for s, e, n in named:
        if last is not None and s > last: gaps.append((s - last, last_name[:60], n[:60], (last - a) / 1e6))
        if last is None or e > last: last, last_name = e, n
    gaps.sort(reverse=True)
    big = [g[0] for g in gaps[:5]]
    where.append(gaps[:3])
    res.append(((b - a) / 1e6, busy / 1e6, len(iv), [round(g / 1e3) for g in big]))
for r in res: print("step %.2f ms  busy %.2f ms  idle %.2f ms  kernels %d  largest gaps (us) %s" % (r[0], r[1], r[0] - r[1], r[2], r[3]))
for g in where[-1]: print("   gap %.0f us at +%.1f ms between [%s] and [%s]" % (g[0] / 1e3, g[3], g[1], g[2]))
PY
