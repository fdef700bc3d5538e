#!/usr/bin/env python3
"""Turn gpurun_out/abl/ (tools/dist_ablation.sh) into profiles/r04_dist_phase_insts.md: per phase of cf_dist_kernel the
instructions per pair emission and the milliseconds it accounts for, by difference between successive ablation builds
(-DCF_DIST_ABL=n removes phases from the end; cf_dist.hip)."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "abl")
E = 151798187198      # pair emissions of the bench's 50 000 reads
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_dist_phase_insts"      # output name under profiles/
names = ["full", "abl1", "abl2", "abl3", "abl4", "abl5", "abl6", "abl7"]
what = {"abl1": "filter + edge rows (hot-list evaluation, totals, row writes, unique bits)", "abl2": "inserts (drains: bucket read, match / claim, count, hot list; overflow list)",
        "abl3": "pushes (ballot, rank, queue store per candidate entry)", "abl4": "table sweep stream + bitmap test (loads, decode, hash, 4 LDS reads, partition / live masks)",
        "abl5": "sketch arithmetic (hash, 4 returning LDS adds, bit-field extract, max test, marks)", "abl6": "sketch sweep stream (item readlanes, loads, decode)",
        "abl7": "clears (sketch 64 KB + bitmap 8 KB + table)"}
rows = {}
for v in names:
    ms = None
    log = os.path.join(src, v + ".log")
    if os.path.exists(log):
        for ln in open(log):
            m = re.search(r"\[([\d.]+), ([\d.]+)", ln)
            if m:
                ms = float(m.group(2))
    c = {}
    for f in glob.glob(os.path.join(src, v, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "cf_dist_kernel" not in r["Kernel_Name"]:
                continue
            c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if ms is not None and c:
        rows[v] = dict(ms=ms, **{k: sum(x) / len(x) for k, x in c.items()})      # mean over the launches (two per run)
if "full" not in rows:
    sys.exit("no counters under gpurun_out/abl")
out = ["# cf_dist_kernel: instructions and time per phase, by ablation", "",
       "`tools/dist_ablation.sh` on an MI355X: `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT` of",
       "`tools/dist_ab.py 50000 <library>` for the shipped library and for builds with `-DCF_DIST_ABL=n` (cf_dist.hip), which remove the kernel's phases from",
       "the END: what runs before the cut is unchanged, so successive builds differ by one phase.  Kernel ms are the tool's HIP-event times WITHOUT the",
       f"profiler's counters; E = {E} pair emissions (the bench's 50 000 reads, edges counted only).", "",
       "LDS busy = SQ_LDS_IDX_ACTIVE / (kernel ms x 2.4 GHz x 256 CUs): the share of the kernel's time a CU's LDS spends on instructions, conflicts included;",
       "conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.", "",
       "| build | kernel ms | VALU / E | SALU / E | LDS / E | LDS busy | conflict |", "|---|---|---|---|---|---|---|"]
for v in names:
    if v in rows:
        r = rows[v]
        out.append(f"| {v} | {r['ms']:.1f} | {r.get('SQ_INSTS_VALU', 0) / E:.3f} | {r.get('SQ_INSTS_SALU', 0) / E:.3f} | {r.get('SQ_INSTS_LDS', 0) / E:.3f} | {r.get('SQ_LDS_IDX_ACTIVE', 0) / (r['ms'] * 1e-3 * 2.4e9 * 256):.2f} | {r.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, r.get('SQ_LDS_IDX_ACTIVE', 0)):.2f} |")
out += ["", "| phase (removed by) | ms | share | VALU / E | SALU / E | LDS / E | LDS cycles / E |", "|---|---|---|---|---|---|---|"]
prev = "full"
full = rows["full"]
for v in names[1:]:
    if v not in rows:
        continue
    a, b = rows[prev], rows[v]
    out.append(f"| {what[v]} ({v}) | {a['ms'] - b['ms']:.1f} | {100 * (a['ms'] - b['ms']) / full['ms']:.0f} % | {(a.get('SQ_INSTS_VALU', 0) - b.get('SQ_INSTS_VALU', 0)) / E:.3f} | "
               f"{(a.get('SQ_INSTS_SALU', 0) - b.get('SQ_INSTS_SALU', 0)) / E:.3f} | {(a.get('SQ_INSTS_LDS', 0) - b.get('SQ_INSTS_LDS', 0)) / E:.3f} | {(a.get('SQ_LDS_IDX_ACTIVE', 0) - b.get('SQ_LDS_IDX_ACTIVE', 0)) / E:.3f} |")
    prev = v
r = rows[prev]
out.append(f"| what is left: loop top, tickets, heads, item records, barriers ({prev}) | {r['ms']:.1f} | {100 * r['ms'] / full['ms']:.0f} % | {r.get('SQ_INSTS_VALU', 0) / E:.3f} | {r.get('SQ_INSTS_SALU', 0) / E:.3f} | {r.get('SQ_INSTS_LDS', 0) / E:.3f} | {r.get('SQ_LDS_IDX_ACTIVE', 0) / E:.3f} |")
clk = full.get("SQ_BUSY_CYCLES")
out += ["", f"Per cycle and CU at the full build ({full['ms']:.1f} ms, 256 CUs, 2.4 GHz): {full.get('SQ_INSTS_VALU', 0) / (full['ms'] * 1e-3 * 2.4e9 * 256):.3f} VALU + "
        f"{full.get('SQ_INSTS_SALU', 0) / (full['ms'] * 1e-3 * 2.4e9 * 256):.3f} SALU wave-instructions.", ""]
open(os.path.join(ROOT, "profiles", tag + ".md"), "w").write("\n".join(out) + "\n")
json.dump(rows, open(os.path.join(ROOT, "profiles", tag + ".json"), "w"), indent=1)
print("\n".join(out))
