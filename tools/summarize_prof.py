#!/usr/bin/env python
"""Copy the judged summaries of a gpurun_out/prof_<tag> run into profiles/<tag>/ (kernel stats + PMC traffic)."""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1]
src, dst = f"gpurun_out/prof_{tag}", f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/trace/trace_kernel_stats.csv", f"{dst}/kernel_stats.csv")
rows = list(csv.DictReader(open(f"{src}/trace/trace_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"{dst}/kernel_stats_top.txt", "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-substep\n")
    f.write("(one launch of every rasteriser kernel covers the 3 views of an event iteration; the single-view launches in\n"
            " the table's min column come from the ground-truth renders bench.py makes before the timed region)\n\n")
    for r in rows[:25]:
        f.write(f"{r['Name'].split('(')[0][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} "
                f"min_us={float(r['MinNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}%\n")
out = {}
for f in ("fetch", "write"):
    p = f"{src}/{f}/{f}_counter_collection.csv"
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    rows_c = list(csv.DictReader(open(p)))
    # keep the launches of the timed steps only (largest grid = all 3 views of an iteration in one launch);
    # bench.py also renders the ground truth with single-view launches before the timed region
    gmax = collections.defaultdict(int)
    for r in rows_c:
        k = r["Kernel_Name"].split("(")[0]
        gmax[k] = max(gmax[k], int(r["Grid_Size"]))
    for r in rows_c:
        k = r["Kernel_Name"].split("(")[0]
        if int(r["Grid_Size"]) == gmax[k]:
            agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[f"{f.upper()}_SIZE_KB_avg"] = sum(v) / len(v)
        out[k][f"{f}_launches"] = len(v)
# gfx950 corrections, calibrated with tools/ubench/pmc_calib.hip on this stack (known byte counts):
#   FETCH_SIZE reports HALF of the fetched bytes for coalesced 16-B/lane streams (1 GiB read -> 524 299 KB) AND for
#   the scattered 48-B record gathers of the compositing kernels (8 Mi records: 654 057 KB = 8 Mi x 1.25 lines x 128 B / 2:
#   it counts 128-B line requests at 64 B) -> x2;  WRITE_SIZE is exact for streams (1 GiB -> 1 048 576 KB) and counts
#   32-B sectors for scattered 48-B stores (8 Mi records: 554 049 KB) -> x1.
FETCH_CORRECTION, WRITE_CORRECTION = 2.0, 1.0
for k, v in out.items():
    v["hbm_bytes_per_launch"] = int((FETCH_CORRECTION * v.get("FETCH_SIZE_KB_avg", 0) +
                                     WRITE_CORRECTION * v.get("WRITE_SIZE_KB_avg", 0)) * 1024)
    v["fetch_correction"], v["write_correction"] = FETCH_CORRECTION, WRITE_CORRECTION
    v["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, KB units, 3-view launches of the timed steps "
                 "only, kernel dispatches serialised by the profiler; hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE "
                 "(gfx950 correction of MI355X_MICROARCH.md, re-calibrated for 48-B record gathers/scatters with "
                 "tools/ubench/pmc_calib.hip: fetches are 128-B lines counted at 64 B)")
# instruction counts of the 3-view launches (profiles/run_pmc_mix.sh <regex> <mixtag>  ->  gpurun_out/pmcm_<mixtag>/a/...)
mixtag = sys.argv[2] if len(sys.argv) > 2 else None
mixcsv = f"gpurun_out/pmcm_{mixtag}/a/a_counter_collection.csv" if mixtag else None
if mixcsv and os.path.exists(mixcsv):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(mixcsv)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] = max(agg[k][r["Counter_Name"]], float(r["Counter_Value"]))
    for k, c in agg.items():
        out.setdefault(k, {})["wave_instructions_per_launch"] = {
            "valu": int(c.get("SQ_INSTS_VALU", 0)), "salu": int(c.get("SQ_INSTS_SALU", 0)), "lds": int(c.get("SQ_INSTS_LDS", 0)),
            "branch": int(c.get("SQ_INSTS_BRANCH", 0)), "vmem": int(c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)),
            "note": "rocprofv3 --pmc SQ_INSTS_* (max over the launches of a profiled bench run = the 3-view launches)"}
json.dump(out, open(f"{dst}/pmc_traffic.json", "w"), indent=1)
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(open(f"{dst}/kernel_stats_top.txt").read())
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "note"} for k, v in out.items()}, indent=1))


# ---- measured traffic per bench stage (tools/pmc_all_traffic.sh -> gpurun_out/traffic_all_<mixtag>.txt): the bench line
# reports it next to the algorithmic bytes of every stage (stages[*].pmc_GB / pmc_GBps)
tall = f"gpurun_out/traffic_all_{mixtag}.txt" if mixtag else None
if tall and os.path.exists(tall):
    shutil.copy(tall, f"{dst}/pmc_traffic_all_kernels.txt")
    per = {}
    for line in open(tall).read().splitlines()[1:]:
        parts = line.rsplit(None, 2)
        if len(parts) == 3:
            try:
                per[parts[0].strip()] = (float(parts[1]) + float(parts[2])) * 1e6        # fetch (corrected) + write, bytes
            except ValueError:
                pass
    def k(prefix):
        return sum(v for name, v in per.items() if name.startswith(prefix) and " @" not in name)
    def depth_sort_bytes():
        """Four three-kernel passes on the 3-view depth keys.  The same kernels also run the (larger) tile sort and the
        single-view launches of the ground-truth renders: per kernel the launch sizes seen are, descending, tile sort
        3 views / 1 view, depth sort 3 views / 1 view -> picked by their workgroup count (onesweep build: the global
        histogram + four onesweep passes instead)."""
        if any(n.startswith("radix_onesweep_kernel") for n in per):
            return k("radix_global_hist_kernel") + 4 * k("radix_onesweep_kernel")
        # launch sizes of the benchmark workload (cfg3: 3 views x 1 M Gaussians = 3 M depth keys, 4096 keys per workgroup;
        # its 256 x 733 counters are scanned by 46 workgroups)
        nb = (3 * 1_000_000 + 4095) // 4096
        want = {"radix_hist_kernel": nb, "radix_scatter_kernel": nb, "void scan_chained_kernel<false>": (256 * nb + 4095) // 4096}
        tot = 0.0
        for kern, g in want.items():
            tot += 4 * per.get("%s @%d" % (kern, g), 0.0)
        return tot
    # one entry = bytes per LAUNCH of the stage as bench.py times it (a stage that is launched twice per iteration is
    # averaged over its two launches, like its avg_ms)
    stage = {
        "preprocess": k("void preprocess_kernel"),
        "sort_depth": depth_sort_bytes(),
        "scan_emit": (k("void bin_kernel<false>") + k("void scan_chained_kernel<true>") + k("void bin_kernel<true>")) / 2,
        "sort_tile": 2 * (k("radix_hist_kernel") + k("void scan_chained_kernel<false>") + k("radix_scatter_kernel")),
        "tile_ranges": (k("tile_ranges_kernel") + 2 * k("tile_order_reg_kernel")) / 2,
        "render_fwd": k("render_fwd_kernel"),
        "render_bwd": k("render_bwd_kernel"),
        "geom_bwd": k("run_reduce_kernel") + k("geom_bwd_multi_kernel"),
        "optimizer": k("sh_adam_views_kernel") + 2 * k("adam_segments_kernel"),
    }
    json.dump({"bytes_per_stage_launch": {a: int(b) for a, b in stage.items()},
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of every kernel (tools/pmc_all_traffic.sh; 2 x FETCH_SIZE + "
                       "WRITE_SIZE, max over launches), summed per bench.py stage"}, open("profiles/traffic_stages.json", "w"), indent=1)
    print(json.dumps(stage, indent=1))
