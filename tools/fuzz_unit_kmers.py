#!/usr/bin/env python3
"""Developer tool (GPU box): randomised differential test of SURVEY §8(f) rank 2 — k-mer OCCURRENCE counts of the de-gapped read rows and the
n most frequent k-mers (reference better_consensus_unit_reconstruction.py:127-135, :156-167) — on random synthetic read sets: k from 5 to 31
(30 is the reference's), n from 0 to more than there are k-mers, both counting paths (sort and reduce in few or many buckets; the atomic
table), against the numpy oracle (oracle/unit_kmers.py).
usage: tools/fuzz_unit_kmers.py [cases] [--seed S] [--seconds T] [--out gpurun_out/fuzz_unit_kmers.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import DeviceError, Engine
from oracle import unit_kmers


def arg(name, default, conv=int):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
seed, budget = arg("--seed", 1), arg("--seconds", 10 ** 9, float)
out = arg("--out", os.path.join(ROOT, "gpurun_out", "fuzz_unit_kmers.json"), str)
rng = np.random.default_rng(seed)
recs, t_start = [], time.time()
lib = None
if os.environ.get("CF_LIB"):
    from centroflye_amd import _lib
    lib = _lib.load(os.environ["CF_LIB"])
with Engine(0, lib) as e:
    for i in range(n_cases):
        if time.time() - t_start > budget:
            break
        n_reads = int(rng.choice([int(x) for x in os.environ.get("CF_FUZZ_READS", "20,100,400,1500").split(",")]))
        sy = dict(seed=int(rng.integers(1, 1 << 30)), n_reads=n_reads, unit_len=int(rng.choice([171, 342, 1026, 2055])), mean_len=float(rng.choice([6000.0, 12000.0, 25000.0])),
                  n_units=int(rng.choice([24, 60, 300])), var_len=int(rng.choice([1, 8])), p_sub=float(rng.uniform(0.0, 0.04)), p_del=float(rng.uniform(0.0, 0.03)), p_ins=float(rng.uniform(0.0, 0.03)))
        k = int(rng.choice([5, 8, 13, 19, 24, 30, 30, 31]))
        knobs = {}
        if rng.random() < 0.5:
            knobs = dict(count_mode=int(rng.choice([0, 1])), count_bits=int(rng.choice([0, 3, 9, 14])))
        rec = dict(case=i, synth=sy, k=k, knobs=knobs)
        t0 = time.time()
        try:
            pk = _host.synth(**sy)
            ro = np.asarray(pk.read_off)
            seqs = [bytes(pk.bases[ro[r]:ro[r + 1]]) for r in range(pk.n_reads)]
            okeys, ocnt = unit_kmers.kmer_occurrences(seqs, k)
            e.set_param("count_mode", 1); e.set_param("count_bits", 0)
            for kk, vv in knobs.items():
                e.set_param(kk, vv)
            e.load(pk, 1); e.count_occurrences(k)
            keys, lo, hi = e.table()
            cnt = lo.astype(np.int64) | (hi.astype(np.int64) << 32)
            ok = bool(np.array_equal(keys, okeys) and np.array_equal(cnt, ocnt))
            tops = []
            for n in (0, 1, int(rng.integers(2, 5000)), okeys.size, okeys.size + 7):
                tk, tc = e.top_kmers(n)
                w = unit_kmers.most_frequent(okeys, ocnt, n)
                tops.append(bool(np.array_equal(tk, okeys[w]) and np.array_equal(tc.astype(np.int64), ocnt[w])))
            rec.update(identical=ok and all(tops), table=ok, tops=tops, n_distinct=int(okeys.size), total=int(ocnt.sum()), max_count=int(ocnt.max()) if ocnt.size else 0)
        except DeviceError as ex:
            rec.update(identical=None if "(-22)" in str(ex) else False, refused=str(ex)[:200])
        except _host.HostError as ex:      # (the generator refuses some draws: no read long enough)
            rec.update(identical=None, refused="generator: " + str(ex)[:120])
        rec["s"] = round(time.time() - t0, 2)
        recs.append(rec)
        if rec["identical"] is False:
            print("DIFFERENCE:", json.dumps(rec), flush=True)
bad = [r for r in recs if r["identical"] is False]
summary = dict(seed=seed, cases=len(recs), identical=sum(1 for r in recs if r["identical"]), refused=sum(1 for r in recs if r["identical"] is None), different=len(bad),
               windows=int(sum(r.get("total", 0) for r in recs)), seconds=round(time.time() - t_start, 1))
json.dump(dict(summary=summary, cases=recs), open(out, "w"), indent=1)
print(json.dumps(summary))
sys.exit(1 if bad else 0)
