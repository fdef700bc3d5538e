#!/usr/bin/env python3
"""bench.py — long-read bases/sec through rare-k-mer recruit + distance (A1-A6) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full pass of the hot path over the synthetic read set, with the packed reads
already resident in HBM: A1 presence table -> A2 rare window + sort -> A3 unit clouds ->
postings -> A5+A6 distance histogram + edge filter.  Workload at N = 1:
BASELINE.json configs[1]/[2] — 50 000 synthetic DXZ1-HOR reads (~1 Gb), k = 19, coverage 32;
for N > 1 every rank holds 50 000 reads of an N-times longer array (weak scaling), counts are
merged with an all-to-all over RCCL, rare lists and clouds are all-gathered, and the distance
stage is partitioned by first k-mer (centroflye_amd/sharded.py).

Rank 0 prints ONE JSON line (contract in the task brief) with two extra objects:
  roofline      the dominant kernel (cf_dist_kernel): algorithmic bytes per launch
                (4 B per pair emission + 4 B per cloud entry + 16 B per stored edge) over its mean
                launch duration, measured with HIP events on the library's own stream
  cpu_baseline  oracle/c (plain-C port of the reference's stage 2) timed on this host on a
                bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import torch  # before libcfhip: both must share one HIP runtime
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from centroflye_amd import _host  # noqa: E402
from centroflye_amd.sharded import ShardedRecruiter  # noqa: E402

K = 19
COVERAGE = 32
PARAMS = dict(k=K, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def synth_kwargs(total_reads, seed):
    # M copies of the 2055-bp unit so that aligned bases / (M * 2055) ~= 32 (reads average ~19.5 kb aligned)
    return dict(seed=seed, n_units=max(24, int(round(0.3 * total_reads))), var_len=8)


def cpu_baseline(sample_reads, seed):
    from oracle import cport
    pk = _host.synth(n_reads=sample_reads, **synth_kwargs(sample_reads, seed))
    up, us, ue, _ = pk.units(1)
    t0 = time.time()
    c, _ = cport.stage2(pk.bases, pk.read_off, up, us, ue, K, PARAMS["max_nonuniq"], PARAMS["lo"], PARAMS["hi"], 0, 2 ** 62,
                        PARAMS["min_d"], PARAMS["max_d"], PARAMS["min_cov"], PARAMS["rel_threshold"])
    dt = time.time() - t0
    return dict(value=pk.n_bases / dt, unit="bases/s", cores=1, kind="port",
                sample=f"{pk.n_reads} reads / {pk.n_bases} bases of the same generator at coverage {COVERAGE} "
                       f"({c['n_emissions']} pair emissions, {dt:.1f} s, oracle/c/cf_oracle.c single thread)",
                host_cpus=os.cpu_count(), emissions_per_s=c["n_emissions"] / dt)


def rr_leg(engine, pk, no_cpu):
    """Read recruitment (reference scripts/read_recruitment/rr.cpp:73-90): every read against the unit and its reverse
    complement, threshold 350; the HOR reads of the workload plus as many random bases that must NOT be recruited."""
    unit = pk.motifs[0].encode()
    rng = np.random.default_rng(3)
    rand = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, pk.n_bases)]
    flat = np.concatenate([pk.bases, rand])
    off = np.concatenate([pk.read_off, pk.read_off[1:] + pk.n_bases])
    for _ in range(2):
        fwd, rc = engine.rr_distances(unit, flat, off, 350)
    ms = engine.times()["rr_kernel_ms"]
    kept = (fwd != -1) | (rc != -1)
    out = {"reads": int(off.size - 1), "bases": int(flat.size), "unit_len": len(unit), "threshold": 350, "kernel_ms": ms,
           "bases_per_s": flat.size / (ms * 1e-3), "recruited": int(kept.sum()), "recruited_hor_reads": int(kept[:pk.n_reads].sum())}
    if not no_cpu:
        from oracle import rr
        sample = list(range(0, pk.n_reads, max(1, pk.n_reads // 12)))[:12] + [pk.n_reads + i for i in range(4)]
        use_ref = rr.ref_distance(b"ACGT", b"ACGT", 1) is not None
        dist = rr.ref_distance if use_ref else rr.distance
        rcu = rr.revcomp(unit)
        t0 = time.time()
        want = [(dist(unit, flat[off[i]:off[i + 1]].tobytes(), 350), dist(rcu, flat[off[i]:off[i + 1]].tobytes(), 350)) for i in sample]
        dt = time.time() - t0
        sb = sum(int(off[i + 1] - off[i]) for i in sample)
        out["cpu_baseline"] = {"value": sb / dt, "unit": "bases/s", "cores": 1, "kind": "reference" if use_ref else "port",
                               "sample": f"{len(sample)} reads / {sb} bases; " + ("the reference's vendored edlib (oracle/_ref)" if use_ref else "oracle/c/cf_oracle_rr.c"),
                               "equal": want == [(int(fwd[i]), int(rc[i])) for i in sample]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50000, help="reads per GPU")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--edge-cap", type=int, default=1 << 26, help="edges stored per GPU (all are counted)")
    ap.add_argument("--cpu-sample-reads", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--place", action="store_true", help="also run stage 3 (A4 + A8/A9 placement) once and report it (N = 1)")
    ap.add_argument("--rr", action="store_true", help="also time read recruitment (SURVEY 8(f) rank 4) on the same reads + as many random ones (N = 1)")
    ap.add_argument("--param", action="append", default=[], help="library knob name=value (cf_set_param)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 or world > 1:
        if world != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} needs WORLD_SIZE={a.gpus} (launch with torch.distributed.run)")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    total_reads = a.reads * world
    t0 = time.time()
    pk = _host.synth(n_reads=a.reads, cand_offset=rank, cand_stride=world, **synth_kwargs(total_reads, a.seed))
    t_synth = time.time() - t0
    sr = ShardedRecruiter(local_rank)
    for p in a.param:
        name, val = p.split("=")
        sr.local.set_param(name, int(val))
        sr.glob.set_param(name, int(val))
    t0 = time.time()
    sr.load(pk, 1)          # reads resident in HBM before the timed region
    t_load = time.time() - t0

    outs = []
    for _ in range(a.warmup):
        outs.append(sr.run(edge_cap=a.edge_cap, **PARAMS))
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    stage_ms = dict(count=0.0, select=0.0, clouds=0.0, postings=0.0, dist=0.0)
    for _ in range(a.steps):
        out = sr.run(edge_cap=a.edge_cap, **PARAMS)
        outs.append(out)
        kernel_ms.append(out["dist_kernel_ms"])
        tl, td = sr.local.times(), sr.dist_engine.times()
        stage_ms["count"] += tl["count_ms"]; stage_ms["select"] += tl["select_ms"]; stage_ms["clouds"] += tl["clouds_ms"]
        stage_ms["postings"] += td["postings_ms"]; stage_ms["dist"] += td["dist_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = outs[-1]
    same = all(all(o[k] == out[k] for k in ("n_edges", "n_emissions", "n_rare", "n_unique", "n_cloud_entries")) for o in outs)
    n_bases = out["n_bases"]
    if rank == 0:
        ms_per_step = elapsed * 1e3 / max(a.steps, 1)
        mean_k_ms = float(np.mean(kernel_ms)) if kernel_ms else 0.0
        # algorithmic bytes of ONE launch of the dominant kernel on this rank:
        # every cloud entry staged once, one 4-byte partner index per pair emission, 16 bytes per stored edge
        alg_bytes = 4 * out["n_cloud_entries"] + 4 * out["local_emissions"] + 16 * min(out["local_edges"], a.edge_cap)
        achieved = alg_bytes / (mean_k_ms * 1e-3) / 1e9 if mean_k_ms > 0 else 0.0
        # HBM traffic of one launch cannot be read from inside this process: it comes from the committed PMC passes
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE with the guide's gfx950 correction) of the same workload, else null
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_dist_kernel.json")) as f:
                pmc = json.load(f)
            if world == 1 and pmc.get("workload_reads_per_gpu") == a.reads:
                traffic, traffic_src = pmc["traffic_bytes_per_launch"], "profiles/r01_pmc_dist_kernel.json"
        except (OSError, KeyError, ValueError):
            pass
        res = {
            "metric": "long-read bases/sec through rare-k-mer recruit+distance",
            "value": n_bases * a.steps / elapsed,
            "unit": "bases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 bases -> u64 k-mers / u32 indices (integer only)",
            "data": "synthetic",
            "config": {"workload": f"{a.reads} synthetic DXZ1-HOR ONT-like reads per GPU ({n_bases} aligned bases in all), "
                                   f"2055-bp unit x {synth_kwargs(total_reads, a.seed)['n_units']} copies, k={K}, coverage {COVERAGE}, "
                                   f"max_distance {PARAMS['max_d']}: count + rare filter + clouds + distance/filter (BASELINE configs[1]+[2], stage 2)",
                       "reads_per_gpu": a.reads, "k": K, "parallelism": f"reads sharded x{world}, first k-mers partitioned x{world}"},
            "roofline": {"bound": "hbm", "kernel": "cf_dist_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch (PMC, see traffic_source)",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms": mean_k_ms, "pair_emissions_per_s": out["local_emissions"] / (mean_k_ms * 1e-3) if mean_k_ms else 0.0},
            "counters": {k: out[k] for k in ("n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_rare", "n_cloud_entries",
                                             "n_emissions", "n_edges", "n_unique", "n_spilled", "n_dist_passes")},
            "stage_ms_per_step": {k: v / max(a.steps, 1) for k, v in stage_ms.items()},
            "setup_s": {"synth": round(t_synth, 2), "load_h2d": round(t_load, 3)},
            "steps_identical": bool(same),
            "device": sr.local.device_info()["name"].strip(),
        }
        if world == 1 and a.place:
            # BASELINE configs[2]: cloud_contig extension on the same reads with the k-mers selected above
            e = sr.local
            t1 = time.perf_counter()
            gk = sr.rare[sr.unique_mask]
            e.set_kmers(gk, K)
            e.build_clouds()
            e.filter_clouds(2)
            cls = pk.classify(50000)
            rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
            t2 = time.perf_counter()
            rd, pos, s0, s1 = e.place_reads(cls, rank, 2, 2, 10, 3)
            t3 = time.perf_counter()
            res["placement"] = {"reads": int(pk.n_reads), "placed": int((pos >= 0).sum()), "none": int((pos < 0).sum()),
                                "classes": np.bincount(cls, minlength=3).tolist(), "clouds_filter_s": t2 - t1, "place_s": t3 - t2,
                                "place_device_ms": e.times()["place_ms"],
                                "end_to_end_bases_per_s": n_bases / (ms_per_step * 1e-3 + (t3 - t1))}
        if world == 1 and a.rr:
            res["read_recruitment"] = rr_leg(sr.local, pk, a.no_cpu_baseline)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.cpu_sample_reads, a.seed)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    sr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
