#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into profiles/:
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats per-kernel table of the default bench run
  <tag>_bench_line.json          the bench line of the same command without the profiler (+ _place_rr: with --place --rr)
  <tag>_pmc_dist_kernel.json     PMC counters of ONE cf_dist_kernel launch + HBM bytes per launch
  <tag>_pmc_other_kernels.json   the same counters for every other kernel of one step (summed over its launches)
usage: tools/pmc_summary.py <tag>"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
for name in ("bench_line", "bench_line_place_rr"):
    path = os.path.join(src, name + ".json")
    if os.path.exists(path):
        lines = [ln for ln in open(path).read().strip().splitlines() if ln.startswith("{")]
        if lines:
            json.loads(lines[-1])
            open(os.path.join(dst, f"{tag}_{name}.json"), "w").write(lines[-1] + "\n")
workload = "bench.py --steps 1 --warmup 0 --no-cpu-baseline --transfer-steps 0 --no-place --edge-cap 3156872828 (50000 reads, the bench's own configuration: all 3 156 871 804 selected edges stored): one launch of every kernel of the step"
per_kernel = {}
for grp in "ABCD":
    for f in glob.glob(os.path.join(src, f"pmc_{grp}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            e = per_kernel.setdefault(k, {}).setdefault(row["Counter_Name"], {"launches": set(), "sum": 0.0})
            e["sum"] += float(row["Counter_Value"]); e["launches"].add(row["Dispatch_Id"])
for k in per_kernel:
    for c in per_kernel[k]:
        per_kernel[k][c] = {"launches": len(per_kernel[k][c]["launches"]), "sum": per_kernel[k][c]["sum"]}
notes = ("rocprofv3 --kernel-trace --pmc <counters>, one pass per counter group (never combined with tracing domains). "
         "FETCH_SIZE/WRITE_SIZE are in KiB; per guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the "
         "bytes of a coalesced stream, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. SQ_ACTIVE_INST_*/SQ_WAVE_CYCLES/SQ_WAIT_* "
         "count quad-cycles summed over waves.")
def traffic(c):
    f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    return (2 * f["sum"] + w["sum"]) * 1024 if f and w else None
dist = [k for k in per_kernel if "cf_dist_kernel" in k]
out = {"workload": workload, "notes": notes, "workload_reads_per_gpu": 50000}
if dist:
    out["kernel"] = dist[0]
    out["counters"] = {c: v["sum"] for c, v in sorted(per_kernel[dist[0]].items())}
    out["dispatches"] = max(v["launches"] for v in per_kernel[dist[0]].values())
    out["traffic_bytes_per_launch"] = traffic(per_kernel[dist[0]])
    # the ratios that DESIGN.md and profiles/README.md quote, written here so that they cannot drift from the counters
    line = os.path.join(dst, f"{tag}_bench_line.json")
    if os.path.exists(line):
        bl = json.loads(open(line).read().strip().splitlines()[-1])
        E = bl["counters"]["n_emissions"]
        c = out["counters"]
        out["per_pair_emission"] = {"n_emissions": E, "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU", 0) / E, "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU", 0) / E,
                                    "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS", 0) / E if "SQ_INSTS_LDS" in c else None}
        out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / bl["roofline"]["algorithmic_bytes_per_launch"] if out["traffic_bytes_per_launch"] else None
        out["traffic_gb_per_launch"] = out["traffic_bytes_per_launch"] / 1e9 if out["traffic_bytes_per_launch"] else None
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_dist_kernel.json"), "w"), indent=1)
others = {k: dict(sorted(v.items()), traffic_bytes=traffic(v)) for k, v in sorted(per_kernel.items()) if k not in dist}
json.dump({"workload": workload, "notes": notes, "kernels": others}, open(os.path.join(dst, f"{tag}_pmc_other_kernels.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "notes"}, indent=1))
for k, v in others.items():
    if v.get("traffic_bytes"):
        print(k, "traffic %.3g GB" % (v["traffic_bytes"] / 1e9), "VALU %.3g" % v.get("SQ_INSTS_VALU", {"sum": 0})["sum"])
