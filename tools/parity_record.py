#!/usr/bin/env python3
"""GPU-box tool: run the ORACLE side of a full-size parity check once on the host's cores and write the record the -m gpu
tests compare the device with (tests/bigparity.py: oracle_record / check_record).  With --gpu the device is checked against
the fresh record in the same process, so a record is only committed after GPU == CPU was seen once.

  python3 tools/parity_record.py --reads 500000 --seed 4 --part 3 --n-parts 8 --sub 16 --through-exchange --gpu \
      --out gpurun_out/r05_parity_500k_rank3.json        # BASELINE configs[3]: one whole rank-of-8 partition (~25 min of 256 threads)
  python3 tools/parity_record.py --reads 200000 --seed 2 --part 21 --n-parts 64 --gpu --out gpurun_out/r05_parity_200k.json

Reference: distance_based_kmer_recruitment.py:39-149 (dist_cnt[d][a] is a's own dict, :108-113: first-k-mer partitions are
independent pieces of the same result)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bigparity

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, required=True); ap.add_argument("--seed", type=int, default=2)
ap.add_argument("--var-len", type=int, default=8); ap.add_argument("--n-units", type=int, default=0)
ap.add_argument("--synth", default="", help="more generator parameters, name=value,name=value")
ap.add_argument("--part", type=int, default=0); ap.add_argument("--n-parts", type=int, default=64); ap.add_argument("--sub", type=int, default=1)
ap.add_argument("--param", action="append", default=[], help="stage-2 parameter name=value (bigparity.P)")
ap.add_argument("--gpu", action="store_true"); ap.add_argument("--through-exchange", action="store_true")
ap.add_argument("--out", required=True)
a = ap.parse_args()
for kv in a.param:
    k, v = kv.split("="); bigparity.P[k] = type(bigparity.P[k])(v)
extra = {}
for kv in filter(None, a.synth.split(",")):
    k, v = kv.split("="); extra[k] = float(v) if "." in v else int(v)
wl = dict(reads=a.reads, seed=a.seed, var_len=a.var_len, n_units=a.n_units or max(24, int(round(0.3 * a.reads))), synth=extra)
t0 = time.time()
pk = bigparity.synth_workload(wl)
print(f"synth: {time.time() - t0:.1f} s, {pk.n_reads} reads, {pk.n_bases} bases", flush=True)
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)


def progress(r):
    with open(a.out + ".partial", "w") as f:
        json.dump(dict(workload=wl, **r), f, indent=1)
    print("progress:", json.dumps(r.get("partial", {k: r[k] for k in ("n_rare", "oracle_A1_A3_s")})), flush=True)


rec = dict(what="oracle/c/cf_oracle_mt.c (OpenMP, all host threads) on the seeded synthetic reads of `workload` — the committed oracle side of tests/test_gpu_fullsize.py",
           workload=wl, host_cpus=os.cpu_count(), **bigparity.oracle_record(pk, a.part, a.n_parts, a.sub, progress))
if a.gpu:
    from centroflye_amd.engine import Engine
    with Engine(0) as e:
        t0 = time.time()
        rec["gpu_at_record_time"] = bigparity.check_record(e, pk, rec, through_exchange=a.through_exchange)
        rec["gpu_at_record_time"]["wall_s"] = round(time.time() - t0, 1)
        rec["gpu_at_record_time"]["device"] = e.device_info()["name"].strip()
with open(a.out, "w") as f:
    json.dump(rec, f, indent=1)
if os.path.exists(a.out + ".partial"):
    os.remove(a.out + ".partial")
print(json.dumps(rec))
sys.exit(0 if (not a.gpu or rec["gpu_at_record_time"]["identical"]) else 1)
