"""Idle time of the queue that carries the step's critical chain, from a rocprofv3 kernel trace of the DEFAULT (overlapped) step.

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/sg -o t -- python $R/bench.py --steps 4 --warmup 2 ... ; python tools/stream_gaps.py /tmp/sg

Per queue: launches, busy time (union of kernel intervals), and for the busiest queue the gaps between one kernel's end and the next one's start
(histogram, and the sum per "previous kernel -> next kernel" pair), so that "launch-boundary time" is a measured number and not 2000 x a guess.
Only the window of the last `--steps` steps is analysed (Adam's launch marks a step's end)."""
import collections
import csv
import glob
import re
import sys


def short(n):
    return re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")[:44]


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r.get("Stream_Id", "0"), short(r["Kernel_Name"]),
                         r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[4].startswith("adam_kernel")]
    if len(ends) < steps + 1:
        print("only %d adam launches in the trace" % len(ends))
        return
    lo, hi = ends[-steps - 1], ends[-1]
    win = rows[lo + 1:hi + 1]
    t0, t1 = win[0][0], max(r[1] for r in win)
    span = (t1 - t0) / 1e6 / steps
    print("window: %d steps, %.2f ms per step, %d launches per step" % (steps, span, len(win) / steps))
    byq = collections.defaultdict(list)
    for r in win:
        byq[(r[2], r[3])].append(r)
    # union of all intervals: time with at least one kernel running
    ev = sorted([(r[0], 1) for r in win] + [(r[1], -1) for r in win])
    busy_any, depth, last = 0, 0, t0
    for t, s in ev:
        if depth > 0:
            busy_any += t - last
        depth += s
        last = t
    print("GPU has >= 1 kernel running: %.2f ms per step; nothing running: %.2f ms per step" % (busy_any / 1e6 / steps, span - busy_any / 1e6 / steps))
    print("%-20s %9s %10s %12s" % ("queue/stream", "launches", "busy ms", "sum dur ms"))
    main_q = None
    for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy, cur_s, cur_e = 0, None, None
        for s, e, *_ in v:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        print("%-20s %9.1f %10.2f %12.2f" % ("%s/%s" % q, len(v) / steps, busy / 1e6 / steps, sum(e - s for s, e, *_ in v) / 1e6 / steps))
        if main_q is None:
            main_q = q
    if len(sys.argv) > 3:                                   # neighbours of the launches whose name matches argv[3], per queue
        pat = re.compile(sys.argv[3])
        ctx = collections.Counter()
        for q, v in byq.items():
            for i, r in enumerate(v):
                if pat.search(r[4]):
                    ctx[("%s/%s" % q, v[i - 1][4] if i else "-", r[4], r[5], v[i + 1][4] if i + 1 < len(v) else "-")] += 1
        print("launches matching %r with their neighbours on the same queue (queue | previous | kernel | grid | next):" % sys.argv[3])
        for k, n in sorted(ctx.items(), key=lambda kv: -kv[1])[:80]:
            print("  %5.1f /step  %-5s %-40s | %-40s %9s | %-40s" % (n / steps, k[0], k[1][:40], k[2][:40], k[3], k[4][:40]))
    v = byq[main_q]
    gaps = []
    for a, b in zip(v, v[1:]):
        gaps.append((b[0] - a[1], a[4], b[4]))
    g = [x[0] for x in gaps]
    print("busiest queue: %d boundaries per step, gap sum %.2f ms per step (positive gaps only: %.2f)" % (len(g) / steps, sum(g) / 1e6 / steps, sum(x for x in g if x > 0) / 1e6 / steps))
    edges = [0, 1000, 2000, 3000, 4000, 6000, 8000, 12000, 20000, 50000, 10 ** 9]
    for lo_, hi_ in zip(edges, edges[1:]):
        sel = [x for x in g if lo_ <= x < hi_]
        print("  gap %5.1f..%7.1f us: %7.1f per step, %6.3f ms per step" % (lo_ / 1e3, hi_ / 1e3, len(sel) / steps, sum(sel) / 1e6 / steps))
    pair = collections.defaultdict(lambda: [0, 0])
    for x, a, b in gaps:
        if x > 0:
            pair[(a, b)][0] += 1
            pair[(a, b)][1] += x
    print("largest boundary sums (prev -> next):")
    for (a, b), (n, s) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:40]:
        print("  %-44s -> %-44s %6.1f /step  avg %5.1f us  %6.3f ms/step" % (a, b, n / steps, s / n / 1e3, s / 1e6 / steps))
    after = collections.defaultdict(lambda: [0, 0])
    for x, a, b in gaps:
        if x > 0:
            after[b][0] += 1
            after[b][1] += x
    print("gap in FRONT of a kernel (by next kernel):")
    for b, (n, s) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %-44s %6.1f /step  avg %5.1f us  %6.3f ms/step" % (b, n / steps, s / n / 1e3, s / 1e6 / steps))


if __name__ == "__main__":
    main()
