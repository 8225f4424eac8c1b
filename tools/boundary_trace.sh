#!/bin/bash
# What does the GPU do between the Adam launch of one step and the first kernels of the next forward pass?
# Kernel + memory-copy trace of a short default bench run; everything (all streams) around the last Adam launch is listed.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/trace_bd
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trace_bd -o t -- python $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline --dcn-offsets 0 > /tmp/trace_bd.log 2>&1
tail -2 /tmp/trace_bd.log | cut -c1-200
python3 - <<'PY'
import csv, glob
WIDE = True
rows = []
for f in glob.glob('/tmp/trace_bd/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q%s ' % r.get('Queue_Id', '?') + r['Kernel_Name'][:70] + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
for f in glob.glob('/tmp/trace_bd/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', '?') + ' bytes=' + r.get('Bytes', r.get('Size', '?'))))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
for which in (adam[-3], adam[-2]):
    nxt = [r for r in rows if r[0] > rows[which][1] + 20e3 and r[2].startswith('K') and ' q2 ' not in r[2]][:6]
    for s, e, n in nxt:
        print("next main-stream kernel: %9.1f us after the Adam launch: %s" % ((s - rows[which][0]) / 1e3, n))
    q2 = [r for r in rows if rows[which][1] < r[0] < nxt[0][0] and ' q2 ' in r[2]]
    print("side-stream kernels in between: %d, last ends %.1f us after the Adam launch (%s)" % (len(q2), (max(r[1] for r in q2) - rows[which][0]) / 1e3, q2[-1][2][:80]))
    t0 = rows[which][0]
    print("---- around Adam at t0")
    for s, e, n in rows:
        if t0 - 150e3 <= s <= rows[which][1] + 4e6 and (n.startswith('C') or (WIDE and ' q2 ' not in n) or s <= rows[which][1] + 0.2e6):
            print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
