#!/usr/bin/env python3
"""Developer tool (GPU box): randomised differential test of stage 2.  Every case draws a synthetic read set (size, unit length,
coverage, read lengths, error rates, divergence, variant model), the script's parameters (k, --max-nonuniq, the rare window,
--min-distance / --max-distance, --min-coverage, the dominance threshold), a first-k-mer partition and — in a third of the cases —
device knobs that force the rarely taken paths (small tables: partition splits and overflow lists; other workgroup shapes; no sketch;
the wide / region layouts; a small hot list; the atomic posting and counting paths), and compares the device with the OpenMP oracle
through tests/bigparity.check: counters, A1 table checksum, rare set, clouds, the partition's emissions / edges / edge checksum /
unique bits.  A quarter of the partitioned cases go the way one RANK of n_parts goes (bigparity.check_record, through_exchange: shard
count, table exchange, gathers and the gathered view through a one-rank communicator that sends to itself, in rounds of a few KB).  A knob combination the library refuses (-22) is recorded as refused; any difference is a failure.
usage: tools/fuzz_parity.py [cases] [--seed S] [--seconds T] [--out gpurun_out/fuzz_parity.json]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import DeviceError, Engine
import bigparity


def arg(name, default, conv=int):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
seed = arg("--seed", 1)
budget = arg("--seconds", 10 ** 9, float)
out = arg("--out", os.path.join(ROOT, "gpurun_out", "fuzz_parity.json"), str)
only = arg("--only", -1)
rng = np.random.default_rng(seed)
BASE = dict(bigparity.P)
KNOB_DEFAULTS = dict(dist_slots=0, dist_block=0, dist_wgs=0, dist_sketch=1, dist_wide=0, dist_regions=0, dist_region_bytes=0, dist_dbits=0, dist_hot_cap=0,
                     dist_post_atomics=0, dist_fill_pct=70, dist_stage=2048, dist_int_thr=1, count_mode=1, count_bits=0, comm_round_bytes=1 << 28,
                     dist_hot_entries=32768, lut_shift=-1)


def draw_case():
    n_reads = int(rng.choice([int(x) for x in os.environ.get("CF_FUZZ_READS", "300,600,1200,2500,5000").split(",")]))
    unit_len = int(rng.choice([342, 1026, 2055, 3078]))
    sy = dict(seed=int(rng.integers(1, 1 << 30)), n_reads=n_reads, unit_len=unit_len, var_len=int(rng.choice([1, 8])),
              mean_len=float(rng.choice([8000.0, 20000.0, 50000.0, 100000.0])),      # (round 6: 100 kb = the cenX shape, ~47 units per read) p_sub=float(rng.uniform(0.003, 0.04)), p_del=float(rng.uniform(0.003, 0.03)),
              p_ins=float(rng.uniform(0.003, 0.03)), unit_div=float(rng.uniform(0.003, 0.03)))
    # (pair emissions grow with the square of the units per read: long reads and short units only on small read sets, so that a case is
    # seconds of oracle time and the run is many cases)
    if sy["mean_len"] > 20000.0:
        n_reads = sy["n_reads"] = min(n_reads, 600 if sy["mean_len"] <= 50000.0 else 300)
    if unit_len < 1000 and sy["mean_len"] > 8000.0:
        unit_len = sy["unit_len"] = 1026
    sy["max_len"] = int(max(200000, 5 * sy["mean_len"]))
    # array units so that the coverage lies between 8 and 60
    cov = float(rng.uniform(8, 60))
    sy["n_units"] = max(24, int(n_reads * sy["mean_len"] / unit_len / cov))
    lo = int(rng.integers(2, 13))
    min_d = int(rng.integers(1, 5))
    p = dict(k=int(rng.choice([9, 11, 15, 19, 19, 19, 23, 27, 31])), max_nonuniq=int(rng.integers(0, 6)), lo=lo, hi=lo + int(rng.integers(3, 60)),
             min_d=min_d, max_d=min_d + int(rng.choice([0, 3, 20, 150, 150, 400])), min_cov=int(rng.choice([1, 2, 3, 4, 4, 4, 6, 9])),
             rel_threshold=float(rng.choice([0.8, 0.8, 0.8, 0.5, 0.6, 0.9, 1.0, 0.3])))
    if n_reads >= 2500:      # (keep the largest read sets away from "every pair is an edge": the oracle's partition and the edge buffer stay small)
        p["lo"] = max(p["lo"], 5); p["hi"] = max(p["hi"], p["lo"] + 3); p["min_cov"] = max(p["min_cov"], 2)
    n_parts = int(rng.choice([1, 1, 2, 3, 7, 16]))
    if n_reads >= 2500 and n_parts == 1:
        n_parts = 4
    if sy["mean_len"] > 20000.0 and n_parts < 7:
        n_parts = 16
    part = int(rng.integers(0, n_parts))
    knobs = {}
    if rng.random() < 0.35:
        for name, choices in (("dist_slots", [256, 512, 2048, 4096]), ("dist_block", [64, 128, 256, 512, 1024]), ("dist_wgs", [1, 2, 3, 4]), ("dist_sketch", [0]),
                              ("dist_wide", [1]), ("dist_regions", [1, 2, 4, 8]), ("dist_region_bytes", [1]), ("dist_dbits", [5, 6, 7, 8]), ("dist_hot_cap", [1, 8, 64]),
                              ("dist_post_atomics", [1]), ("dist_fill_pct", [20, 50, 90]), ("dist_stage", [0, 64]), ("dist_int_thr", [0]), ("count_mode", [0]),
                              ("count_bits", [4, 9, 14]), ("dist_hot_entries", [-1, 0, 2000]), ("lut_shift", [0, 1, 3])):
            if rng.random() < 0.18:
                knobs[name] = int(rng.choice(choices))
    return sy, p, part, n_parts, knobs


recs, t_start = [], time.time()
lib = None
if os.environ.get("CF_LIB"):      # (another build of the device library, e.g. the host emulator for a dry run of this script)
    from centroflye_amd import _lib
    lib = _lib.load(os.environ["CF_LIB"])
with Engine(0, lib) as e:
    for i in range(n_cases):
        if time.time() - t_start > budget:
            break
        sy, p, part, n_parts, knobs = draw_case()
        exchange = bool(rng.random() < 0.25) and n_parts >= 2
        rec = dict(case=i, synth=sy, params=p, partition=[part, n_parts], knobs=knobs, exchange=exchange)
        if only >= 0 and i != only:      # (--only i: the i-th case of this seed alone)
            continue
        t0 = time.time()
        try:
            pk = _host.synth(**sy)
            bigparity.P.clear(); bigparity.P.update(BASE); bigparity.P.update(p)
            for kk, vv in KNOB_DEFAULTS.items():
                e.set_param(kk, vv)
            for kk, vv in knobs.items():
                e.set_param(kk, vv)
            # (a look at the size first, on the device: a case whose partition has more than 4e9 pair emissions or 2e8 edges is minutes of oracle time
            # and gigabytes of edge rows — dropped, not run)
            e.load(pk, 1); e.count_kmers(p["k"]); e.select_rare(p["max_nonuniq"], p["lo"], p["hi"]); e.build_clouds(); e.reset_unique()
            ne0 = e.dist_edges(0, 2 ** 62, p["min_d"], p["max_d"], p["min_cov"], p["rel_threshold"], part, n_parts, edge_cap=0)
            if e.stats()["n_emissions"] > 4e9 or ne0 > 2e8:
                raise DeviceError(f"case too large (-12): {e.stats()['n_emissions']} pair emissions, {ne0} edges")
            if exchange:
                # the way ONE rank of n_parts runs its partition: A1 on its read shard, table exchange / rare gather / cloud gather through a one-rank
                # communicator that sends to itself (bucketing, rounds of comm_round_bytes, merge, gathered view), A5 / A6 over the gathered view
                e.set_param("comm_round_bytes", int(rng.choice([1 << 12, 1 << 16, 1 << 28])))
                orec = bigparity.oracle_record(pk, part, n_parts)
                x = bigparity.check_record(e, pk, orec, through_exchange=True, rendezvous=tempfile.mkdtemp() if lib else None)      # (the emulator's file transport meets in a directory)
                e.set_param("comm_round_bytes", 1 << 28)
                r = dict(identical=x["identical"], checks=x["checks"], n_rare=orec["n_rare"], n_emissions_partition=x["got"]["n_emissions_partition"],
                         n_edges_partition=x["got"]["n_edges_partition"], n_dist_passes=x["got"]["n_dist_passes"], n_bases=orec["n_bases"],
                         dist_kernel_ms=x["got"]["dist_kernel_ms"], oracle_A1_A3_s=orec["oracle_A1_A3_s"], oracle_partition_s=orec["partition"]["oracle_s"])
            else:
                r = bigparity.check(e, pk, part=part, n_parts=n_parts)
            rec.update(identical=bool(r["identical"]), checks=r["checks"], n_rare=r["n_rare"], n_emissions=r["n_emissions_partition"], n_edges=r["n_edges_partition"],
                       passes=r["n_dist_passes"], n_bases=r["n_bases"], dist_kernel_ms=r["dist_kernel_ms"], oracle_s=round(r["oracle_A1_A3_s"] + r["oracle_partition_s"], 1),
                       device_ms={k: round(float(v), 1) for k, v in e.times().items() if k.endswith("_ms") and v})
        except DeviceError as ex:
            refused = "(-22)" in str(ex) or "(-12)" in str(ex)      # (a knob combination the library does not take; a case too large for the device)
            rec.update(identical=None if refused else False, refused=str(ex)[:200])
        rec["s"] = round(time.time() - t0, 2)
        recs.append(rec)
        print(json.dumps({k: rec.get(k) for k in ("case", "identical", "refused", "params", "partition", "exchange", "knobs", "n_rare", "n_emissions", "n_edges", "passes", "dist_kernel_ms", "oracle_s", "s")}), flush=True)
        if rec["identical"] is False:
            print("DIFFERENCE:", json.dumps(rec), flush=True)
bad = [r for r in recs if r["identical"] is False]
summary = dict(seed=seed, cases=len(recs), identical=sum(1 for r in recs if r["identical"]), refused=sum(1 for r in recs if r["identical"] is None), different=len(bad),
               with_edges=sum(1 for r in recs if r.get("n_edges")), through_exchange=sum(1 for r in recs if r.get("exchange") and r["identical"]), pair_emissions=int(sum(r.get("n_emissions") or 0 for r in recs)), seconds=round(time.time() - t_start, 1))
json.dump(dict(summary=summary, cases=recs), open(out, "w"), indent=1)
print(json.dumps(summary))
sys.exit(1 if bad else 0)
