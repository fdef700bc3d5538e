#!/usr/bin/env python3
"""Developer tool (GPU box): randomised differential test of stage 3.  Every case draws a synthetic read set (size, coverage from thick
to so thin that most reads stay unplaced, read lengths, error rates), takes the unique k-mers of the device's own stage 2, draws the
placer's parameters (--min-kmer-mult, --min-cloud-kmer-freq, --min-unit, --min-inters, --prefix-threshold) and, in half of the cases,
device knobs (launch shapes of the iteration kernel, posting-row width, score regions that start too small, the third level of the
arg-max, the hash-map path), and compares every line of read_positions.csv with the C placer (oracle/c/cf_oracle_place.c).
usage: tools/fuzz_place.py [cases] [--seed S] [--seconds T] [--out gpurun_out/fuzz_place.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import DeviceError, Engine
from oracle import cport
from conftest import lines_from_placement


def arg(name, default, conv=int):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
seed = arg("--seed", 1)
budget = arg("--seconds", 10 ** 9, float)
out = arg("--out", os.path.join(ROOT, "gpurun_out", "fuzz_place.json"), str)
only = arg("--only", -1)
rng = np.random.default_rng(seed)
KNOB_DEFAULTS = dict(place_mode=2, place_grid=0, place_block=0, place_row_words=0, place_slots_per_unit=0, place_fused=1, place_l3=0, place_l3_shift=0)

recs, t_start = [], time.time()
lib = None
if os.environ.get("CF_LIB"):      # (another build of the device library, e.g. the host emulator for a dry run of this script)
    from centroflye_amd import _lib
    lib = _lib.load(os.environ["CF_LIB"])
with Engine(0, lib) as e:
    for i in range(n_cases):
        if time.time() - t_start > budget:
            break
        n_reads = int(rng.choice([int(x) for x in os.environ.get("CF_FUZZ_READS", "200,500,1000,2000,4000").split(",")]))
        mean_len = float(rng.choice([8000.0, 20000.0, 40000.0]))
        upr = float(rng.choice([0.15, 0.3, 0.3, 0.6, 1.5]))      # array units per read: 0.3 = coverage 32 at 20 kb reads; 1.5: thin, most reads unplaced
        sy = dict(seed=int(rng.integers(1, 1 << 30)), n_reads=n_reads, n_units=max(24, int(round(upr * n_reads * mean_len / 20000.0))), var_len=int(rng.choice([1, 8])),
                  mean_len=mean_len, p_sub=float(rng.uniform(0.005, 0.03)), p_del=float(rng.uniform(0.005, 0.03)), p_ins=float(rng.uniform(0.005, 0.025)))
        s = dict(min_mult=int(rng.choice([1, 2, 2, 3])), freq=int(rng.choice([1, 2, 2, 2, 3, 4])), min_unit=int(rng.choice([1, 2, 2, 3, 4])),
                 min_inters=int(rng.choice([1, 3, 4, 10, 10, 10, 30, 80])), prefix_threshold=int(rng.choice([50000, 50000, 5000, 200000])))
        knobs = {}
        if rng.random() < 0.5:
            for name, choices in (("place_grid", [1, 3, 16, 64, 200]), ("place_block", [256, 512]), ("place_row_words", [64]), ("place_slots_per_unit", [1, 4, 16]),
                                  ("place_l3", [1]), ("place_l3_shift", [1, 2, 3]), ("place_mode", [1, 3]), ("place_fused", [0])):
                if rng.random() < 0.25:
                    knobs[name] = int(rng.choice(choices))
        for kv in filter(None, os.environ.get("CF_FUZZ_KNOBS", "").split(",")):      # (knobs forced on every case, e.g. place_mode=1: one case again on the other path)
            knobs[kv.split("=")[0]] = int(kv.split("=")[1])
        n_motif = int(rng.choice([1, 1, 1, 2]))      # (--n-motif: units of n motif copies, ncrf_parser.py:28-59)
        rec = dict(case=i, synth=sy, params=s, knobs=knobs, n_motif=n_motif)
        if only >= 0 and i != only:      # (--only i: the i-th case of this seed alone; the draws before it are made and dropped)
            continue
        t0 = time.time()
        try:
            pk = _host.synth(keep_rows=n_motif != 1, **sy)      # (units of n > 1 motif copies are cut from the alignment rows)
            cls = pk.classify(s["prefix_threshold"])
            rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
            up = pk.units(n_motif)[0]
            for kk, vv in KNOB_DEFAULTS.items():
                e.set_param(kk, vv)
            e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds(); e.reset_unique()
            e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
            gk = e.kmers()[e.unique_mask()]
            if n_motif != 1:
                e.load(pk, n_motif)
            rec.update(n_kmers=int(gk.size), classes=[int((cls == c).sum()) for c in range(3)])
            if gk.size == 0:
                rec.update(identical=None, refused="no unique k-mers")
            else:
                e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(s["min_mult"])
                cp, ent = e.clouds()
                rec["n_entries"] = int(ent.size)
                tc = time.time()
                want = lines_from_placement(pk.ids, *[x.tolist() for x in cport.place_reads(cls, rank, up, cp, ent, gk.size, s["freq"], s["min_unit"], s["min_inters"], 3)])
                rec["oracle_s"] = round(time.time() - tc, 2)
                for kk, vv in knobs.items():
                    e.set_param(kk, vv)
                got = lines_from_placement(pk.ids, *[x.tolist() for x in e.place_reads(cls, rank, s["freq"], s["min_unit"], s["min_inters"], 3)])
                rec.update(identical=got == want, placed=sum(1 for x in want if not x.endswith("None")), place_ms=round(float(e.times()["place_ms"]), 1))
                if got != want:
                    rec["first_difference"] = next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want)))
        except DeviceError as ex:
            refused = "(-22)" in str(ex) or "(-12)" in str(ex) or ("(-34)" in str(ex) and knobs.get("place_mode") == 3)      # (mode 3 never falls back: it may refuse)
            rec.update(identical=None if refused else False, refused=str(ex)[:200])
        rec["s"] = round(time.time() - t0, 2)
        recs.append(rec)
        print(json.dumps({k: rec.get(k) for k in ("case", "identical", "refused", "params", "n_motif", "knobs", "n_kmers", "n_entries", "classes", "placed", "place_ms", "oracle_s", "s")}), flush=True)
        if rec["identical"] is False:
            print("DIFFERENCE:", json.dumps(rec), flush=True)
bad = [r for r in recs if r["identical"] is False]
summary = dict(seed=seed, cases=len(recs), identical=sum(1 for r in recs if r["identical"]), refused=sum(1 for r in recs if r["identical"] is None), different=len(bad),
               placed=int(sum(r.get("placed") or 0 for r in recs)), seconds=round(time.time() - t_start, 1))
json.dump(dict(summary=summary, cases=recs), open(out, "w"), indent=1)
print(json.dumps(summary))
sys.exit(1 if bad else 0)
