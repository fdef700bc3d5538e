#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into profiles/:
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats per-kernel table of the default bench run
  <tag>_bench_line.json          the bench line of the same command without the profiler
  <tag>_pmc_dist_kernel.json     PMC counters of ONE cf_dist_kernel launch + HBM bytes per launch
usage: tools/pmc_summary.py <tag>"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
line = open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line + "\n")
out = {}
workload = "bench.py --steps 1 --warmup 0 --no-cpu-baseline (50000 reads), the single cf_dist_kernel<cf_tab_narrow> launch"
for grp in "ABCD":
    acc, n_disp = {}, set()
    for f in glob.glob(os.path.join(src, f"pmc_{grp}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "cf_dist_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                n_disp.add(row["Dispatch_Id"])
    out[grp] = {"workload": workload, "dispatches": len(n_disp), "counters": dict(sorted(acc.items()))}
fetch = out["C"]["counters"].get("FETCH_SIZE"); write = out["D"]["counters"].get("WRITE_SIZE")
out["notes"] = ("rocprofv3 --kernel-trace --pmc <counters>, one pass per counter group (never combined with tracing domains). "
                "FETCH_SIZE/WRITE_SIZE are in KiB; per guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the "
                "bytes of a coalesced stream, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. SQ_ACTIVE_INST_*/SQ_WAVE_CYCLES/SQ_WAIT_* "
                "count quad-cycles summed over waves.")
if fetch is not None and write is not None:
    out["traffic_bytes_per_launch"] = (2 * fetch + write) * 1024
out["workload_reads_per_gpu"] = 50000
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_dist_kernel.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in "ABCD"}, indent=1))
for grp in "ABCD":
    print(grp, out[grp]["dispatches"], out[grp]["counters"])
