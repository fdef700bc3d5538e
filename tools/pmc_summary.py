#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into profiles/:
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats per-kernel table of the default bench run
  <tag>_bench_line.json          the bench line of the same command without the profiler (+ _place_rr: with --place --rr)
  <tag>_pmc_dist_kernel.json     PMC counters of ONE cf_dist_kernel launch + HBM bytes per launch
  <tag>_pmc_other_kernels.json   the same counters for every other kernel of one step (summed over its launches)
usage: tools/pmc_summary.py <tag>"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
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
workload = "bench.py --steps 1 --warmup 0 --no-cpu-baseline --transfer-steps 0 --no-place --steps-b 0 --steps-c 0 --edge-cap 3156872828 (50000 reads, the bench's own configuration: all 3 156 871 804 selected edges stored): one launch of every kernel of the step"
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
# The guide's doubling of FETCH_SIZE is stated for WIDE COALESCED STREAMING reads (>= 16 B per lane, contiguous: 128-byte requests
# tallied at 64 B).  Kernels whose HBM-side reads are mostly random 64-byte requests (one lookup slot, one posting, one item record per
# lane) are not that pattern: for them the raw figure is probably exact and the doubled one an upper bound — both are written
# (VERDICT round 4).  The list names the kernels of this library by what their dominant loads look like in the source.
RANDOM_REQUEST_KERNELS = ("cf_cloud_kernel", "cf_items_count_kernel", "cf_items_fill_kernel", "cf_sum_partner_kernel", "cf_post_fill_kernel",
                          "cf_lut_build_kernel", "cf_unit_rend_kernel", "cf_pl2_", "cf_place_")
def is_streaming(kernel):
    return not any(kernel.startswith(p) or (" " + p) in kernel for p in RANDOM_REQUEST_KERNELS)
def traffic(c, streaming=True):
    f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    return ((2 if streaming else 1) * f["sum"] + w["sum"]) * 1024 if f and w else None
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
        # what binds the kernel, from these counters and the kernel time of the same configuration's bench line (bench.py copies this
        # block into roofline.lds / l2_hit / hbm_side_gbps / bound): per CU-cycle figures use 256 CUs at 2.4 GHz
        k_ms = bl["roofline"]["kernel_ms"]
        cu_cycles = k_ms * 1e-3 * 2.4e9 * 256
        lds_active = c.get("SQ_LDS_IDX_ACTIVE", 0) / cu_cycles
        valu, salu, ldsi = c.get("SQ_INSTS_VALU", 0) / cu_cycles, c.get("SQ_INSTS_SALU", 0) / cu_cycles, c.get("SQ_INSTS_LDS", 0) / cu_cycles
        hbm_gbps = out["traffic_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 if out["traffic_bytes_per_launch"] else None
        wait = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
        d = {"kernel_ms": k_ms,
             "lds": {"idx_active_frac": round(lds_active, 4),
                     "bank_conflict_frac": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4) if c.get("SQ_LDS_IDX_ACTIVE") else None,
                     "insts_per_pair": round(c.get("SQ_INSTS_LDS", 0) / E, 4), "waves_per_simd": 4},
             "issue_per_cycle_per_cu": {"valu": round(valu, 3), "salu": round(salu, 3), "lds": round(ldsi, 3)},
             "wait_any_frac_of_wave_cycles": round(wait, 3) if wait is not None else None,
             "l2_hit": round(c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"]), 4) if c.get("TCC_HIT") else None,
             "hbm_side_gbps": round(hbm_gbps, 1) if hbm_gbps else None, "hbm_side_frac_of_peak": round(hbm_gbps / 8000.0, 4) if hbm_gbps else None}
        # the label: HBM-bound only if the HBM side itself is near what the memory sustains (6.3 of 8 TB/s achievable); otherwise the
        # busiest on-chip resource — here no unit is saturated and waves wait on returning LDS operations at 4 waves per SIMD
        if hbm_gbps and hbm_gbps > 0.6 * 6300:
            d["bound"] = "hbm"
        elif lds_active > 0.35 and wait is not None and wait > 0.4 and max(valu, salu) < 0.8:
            d["bound"] = "lds-latency"
        elif valu + salu > 1.0:
            d["bound"] = "issue"
        else:
            d["bound"] = "latency"
        out["derived"] = d
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_dist_kernel.json"), "w"), indent=1)
others = {k: dict(sorted(v.items()), traffic_bytes=traffic(v, is_streaming(k)), traffic_bytes_fetch_doubled=traffic(v, True), traffic_bytes_fetch_raw=traffic(v, False),
                   loads="streaming (FETCH_SIZE doubled)" if is_streaming(k) else "random 64-byte requests (FETCH_SIZE raw; doubled = upper bound)")
          for k, v in sorted(per_kernel.items()) if k not in dist}
json.dump({"workload": workload, "notes": notes, "kernels": others}, open(os.path.join(dst, f"{tag}_pmc_other_kernels.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "notes"}, indent=1))
for k, v in others.items():
    if v.get("traffic_bytes"):
        print(k, "traffic %.3g GB" % (v["traffic_bytes"] / 1e9), "VALU %.3g" % v.get("SQ_INSTS_VALU", {"sum": 0})["sum"])


# ---- round 6: the same counters of cf_dist_kernel on the cenX-shaped reads of bench.py's workload_c (pmcc_* passes of tools/profile_round.sh)
pk_c = {}
for grp in "ABCD":
    for f in glob.glob(os.path.join(src, f"pmcc_{grp}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "cf_dist_kernel" not in k:
                continue
            e = pk_c.setdefault(row["Counter_Name"], {"launches": set(), "sum": 0.0})
            e["sum"] += float(row["Counter_Value"]); e["launches"].add(row["Dispatch_Id"])
line = os.path.join(dst, f"{tag}_bench_line.json")
if pk_c and os.path.exists(line):
    bl = json.loads(open(line).read().strip().splitlines()[-1])
    wc = (bl.get("workload_c") or {}).get("var_len_8")
    c = {k: v["sum"] for k, v in sorted(pk_c.items())}
    oc = {"workload": "tools/cenx_probe.py --var-len 8 --once 1809975565: ONE launch of cf_dist_kernel on the cenX-shaped reads of bench.py's workload_c (1 000 reads of mean 100 kb over a 1 500-unit array, coverage 32: "
                      "3.217e10 pair emissions, ~60 800 per first k-mer, all 1 809 974 541 selected edges stored), launch shape 1 x 1024 threads per CU",
          "notes": notes, "kernel": "cf_dist_kernel<cf_tab_narrow_t<8, 4>>", "counters": c, "dispatches": max(len(v["launches"]) for v in pk_c.values())}
    tb = ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024) if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None
    oc["traffic_bytes_per_launch"] = tb
    if wc:
        E, k_ms, alg = wc["counters"]["n_emissions"], wc["roofline"]["kernel_ms"], wc["roofline"]["algorithmic_bytes_per_launch"]
        cu_cycles = k_ms * 1e-3 * 2.4e9 * 256
        oc["per_pair_emission"] = {"n_emissions": E, "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU", 0) / E, "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU", 0) / E, "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS", 0) / E}
        oc["traffic_over_algorithmic"] = tb / alg if tb else None
        oc["derived"] = {"kernel_ms_of_the_bench_line": k_ms, "algorithmic_bytes_per_launch": alg, "roofline_frac": alg / (k_ms * 1e-3) / 8e12,
                         "pair_emissions_per_s": E / (k_ms * 1e-3), "dist_passes_per_first_kmer": wc.get("dist_passes_per_first_kmer"),
                         "lds": {"idx_active_frac": round(c.get("SQ_LDS_IDX_ACTIVE", 0) / cu_cycles, 4),
                                 "bank_conflict_frac": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4) if c.get("SQ_LDS_IDX_ACTIVE") else None,
                                 "insts_per_pair": round(c.get("SQ_INSTS_LDS", 0) / E, 4), "waves_per_simd": 4},
                         "issue_per_cycle_per_cu": {"valu": round(c.get("SQ_INSTS_VALU", 0) / cu_cycles, 3), "salu": round(c.get("SQ_INSTS_SALU", 0) / cu_cycles, 3), "lds": round(c.get("SQ_INSTS_LDS", 0) / cu_cycles, 3)},
                         "wait_any_frac_of_wave_cycles": round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3) if c.get("SQ_WAVE_CYCLES") else None,
                         "l2_hit": round(c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"]), 4) if c.get("TCC_HIT") else None,
                         "hbm_side_gbps": round(tb / (k_ms * 1e-3) / 1e9, 1) if tb else None}
    json.dump(oc, open(os.path.join(dst, f"{tag}_pmc_dist_kernel_workload_c.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in oc.items() if k != "notes"}, indent=1))
