#!/usr/bin/env python
"""Kernel timeline of one training step from a rocprofv3 kernel-trace CSV: start, duration, gap to the previous kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:40] for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith("adam_segments_kernel")]     # one launch ends an iteration
ends = adam
a, b = ends[-3] + 1, ends[-2] + 1
seg = rows[a:b]
t0 = int(rows[a - 1]["End_Timestamp"])
prev, busy, gaps = t0, 0.0, 0.0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f gap %6.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].split("(")[0][:34]))
    busy += (e - s) / 1e3; gaps += max(0, s - prev) / 1e3; prev = e
print("step span %.1f us, busy %.1f us, gaps %.1f us" % ((prev - t0) / 1e3, busy, gaps))
