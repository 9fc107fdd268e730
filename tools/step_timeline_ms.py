#!/usr/bin/env python
"""Multi-stream step timeline from a rocprofv3 kernel trace CSV: union busy time, per-kernel totals, big-kernel schedule."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:40] for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
ends = adam[5::6]   # 6 adam launches per step (5 groups + c)
a, b = ends[2] + 1, ends[3] + 1
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
print("step span ms", (t1 - t0) / 1e6, "kernels", len(seg))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("union busy ms", busy / 1e6, "idle ms", (t1 - t0 - busy) / 1e6)
agg = collections.OrderedDict()
for r in seg:
    n = r["Kernel_Name"].split("(")[0][:50]
    d = agg.setdefault(n, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{n:50s} x{c:3d} {t/1e3:9.1f} us")
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d > 60000:
        print("%8.1f +%7.1f us  q%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d / 1e3, r["Queue_Id"], r["Kernel_Name"].split("(")[0][:30]))
