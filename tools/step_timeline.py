#!/usr/bin/env python
"""Summarise one training step from a rocprofv3 kernel trace CSV: per-kernel time and idle gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:48] for r in rows]
# find step boundaries: adam_kernel groups (5 consecutive per step)
adam = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] != adam[i] + 1 and (i + 1) % 5 == 0]
ends = adam[4::5]
if len(ends) < 3:
    print("not enough steps"); sys.exit()
a, b = ends[-3] + 1, ends[-2] + 1           # one full step
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"step span {1e-6*(t1-t0):.3f} ms, kernels {len(seg)}, busy {1e-6*busy:.3f} ms, idle {1e-6*(t1-t0-busy):.3f} ms")
agg = collections.OrderedDict()
for r in seg:
    n = r["Kernel_Name"].split("(")[0][:60]
    d = agg.setdefault(n, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:60s} x{c:3d} {1e-3*t:9.1f} us")
gaps = []
for p, q in zip(seg[:-1], seg[1:]):
    g = int(q["Start_Timestamp"]) - int(p["End_Timestamp"])
    if g > 15000:
        gaps.append((g, p["Kernel_Name"].split("(")[0][:40], q["Kernel_Name"].split("(")[0][:40]))
print("gaps > 15 us:")
for g, p, q in sorted(gaps, reverse=True)[:15]:
    print(f"  {1e-3*g:8.1f} us  after {p} before {q}")
