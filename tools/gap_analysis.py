#!/usr/bin/env python
"""GPU idle-gap analysis of a rocprofv3 --kernel-trace CSV: union of kernel intervals vs wall time over the steady-state
steps, and the kernels that follow the largest gaps (= where the GPU waited for the host or for a dependency)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady state: last 60 % of the trace
t0 = iv[0][0] + int(0.4 * (iv[-1][1] - iv[0][0]))
iv = [v for v in iv if v[0] >= t0]
wall = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.1f ms, busy %.1f ms (%.1f %%), idle %.1f ms in %d gaps" % (wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, len(gaps)))
hist = {}
for g, n in gaps:
    b = "<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"
    h = hist.setdefault(b, [0, 0])
    h[0] += 1
    h[1] += g
for b in ("<5us", "<20us", "<100us", "<1ms", ">=1ms"):
    if b in hist:
        print("  gaps %-7s n=%6d total %.2f ms" % (b, hist[b][0], hist[b][1] / 1e6))
after = {}
for g, n in gaps:
    k = n[:70]
    a = after.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += g
print("largest idle time by the kernel that ends the gap:")
for k, (c, t) in sorted(after.items(), key=lambda x: -x[1][1])[:25]:
    print("  %8.2f ms %6d  %s" % (t / 1e6, c, k))
