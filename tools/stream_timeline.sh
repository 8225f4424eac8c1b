#!/bin/bash
# Which stream is the critical path of a training step?  rocprofv3 --kernel-trace of the default (overlapped) bench run; per step
# (delimited by the fused-Adam launches) and per HSA queue: first start, last end, busy time -- and how long the weight-gradient
# queue keeps running after the main queue's last backward kernel.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/trace_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_tl -o t -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --dcn-offsets 0 > /tmp/trace_tl.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/trace_tl/*kernel_trace.csv')[0]
rd = csv.DictReader(open(f))
print("columns:", rd.fieldnames)
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rd]
rows.sort()
adam = [s for s, e, n, q, st in rows if 'adam_kernel' in n]          # one launch per step (prn_adam_step)
marks = adam
for a, b in list(zip(marks[5:-1], marks[6:]))[:4]:
    iv = [r for r in rows if a <= r[0] < b]
    perq = collections.OrderedDict()
    for s, e, n, q, st in iv:
        d = perq.setdefault((q, st), [s, e, 0, 0, n])
        d[1] = max(d[1], e); d[2] += e - s; d[3] += 1
    print("step %.2f ms, %d kernels" % ((b - a) / 1e6, len(iv)))
    for (q, st), (s, e, busy, cnt, first) in perq.items():
        print("   queue %s stream %s: %5d kernels, from +%.2f to +%.2f ms, busy %.2f ms   (first: %s)" % (q, st, cnt, (s - a) / 1e6, (e - a) / 1e6, busy / 1e6, first[:50]))
    # last kernel per queue before the next Adam
    wg = [r for r in iv if 'conv_wgrad_kernel' in r[2] or 'reduce_splits' in r[2] or 'winograd_dw' in r[2]]
    if wg:
        qs = collections.Counter(r[3] for r in wg).most_common(1)[0][0]
        last_side = max(r[1] for r in iv if r[3] == qs)
        main_q = collections.Counter(r[3] for r in iv if 'bn_' in r[2]).most_common(1)[0][0]
        bwd_main = [r for r in iv if r[3] == main_q]
        # the main queue's last kernel that is not the optimizer / loss of the next step: take the last 'conv_igemm' or bn_bwd
        last_bwd = max(r[1] for r in bwd_main if ('bn_bwd' in r[2] or 'bn_small_bwd' in r[2] or 'conv_igemm' in r[2] or 'maxpool_bwd' in r[2]))
        print("   weight-gradient queue %s ends +%.2f ms; main queue %s: last backward kernel ends +%.2f ms -> tail of %.2f ms" % (qs, (last_side - a) / 1e6, main_q, (last_bwd - a) / 1e6, (last_side - last_bwd) / 1e6))
PY
