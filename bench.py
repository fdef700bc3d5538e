#!/usr/bin/env python3
"""bench.py — long-read bases/sec through rare-k-mer recruit + distance (A1-A6) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full pass of the hot path over the synthetic read set, with the packed reads
already resident in HBM: A1 presence table -> A2 rare window + sort -> A3 unit clouds ->
postings -> A5+A6 distance histogram + edge filter, every selected edge stored.
Workload at N = 1: BASELINE.json configs[1]/[2] — 50 000 synthetic DXZ1-HOR reads (~1 Gb),
k = 19, coverage 32; for N > 1 every rank holds 50 000 reads of an N-times longer array (weak
scaling): counts are merged with an all-to-all over RCCL, rare lists and clouds are all-gathered,
the distance stage is partitioned by first k-mer (centroflye_amd/sharded.py; all of it inside
libcfhip.so — this process never imports torch).

N > 1: either launched once per rank by `python -m torch.distributed.run` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_PORT read from the environment; the RCCL unique id travels through a file named after
the launcher's pid), or started plainly as `python bench.py --gpus N`, in which case this process
— which never touches a GPU — starts the N ranks as child processes itself.

Rank 0 prints ONE JSON line (contract in the task brief) with these extra objects:
  roofline              the dominant kernel (cf_dist_kernel): algorithmic bytes per launch (4 B per pair
                        emission + 4 B per cloud entry + 16 B per stored edge) over its mean launch
                        duration, measured with HIP events on the library's own stream; plus
                        whole_step_frac = SURVEY §8(d)'s B_alg of the whole step / ms_per_step / 8 TB/s
  value_incl_transfers  the same step with the H2D of the packed reads and the D2H of the rare set,
                        the unique mask and stored edges inside the timed region
  cpu_baseline          oracle/c (plain-C port of the reference's stage 2) timed on this host on the
                        benchmark's own reads: A1-A3 whole + A5/A6 for one first-k-mer partition of 64
                        (x 64), all cores (OpenMP); second leg: the partitioned A5/A6 on one thread
  parity_vs_committed_oracle   counters + device-side checksum of every stored edge against the 64/64
                        CPU run of the oracle on the same reads (profiles/r03_full_parity.json)
  end_to_end            stage 2 + A4 + greedy placement of the same reads (BASELINE configs[2])
  workload_c            (N = 1) BASELINE configs[4]'s SHAPE: 1 000 reads of mean 100 kb over a 1 500-unit array at coverage 32 (~47 units per read,
                        ~60 000 pair emissions per first k-mer), var_len 8 and 1: ms_per_step, stage split, passes per first k-mer, pair emissions/s,
                        roofline figures, and every figure of the step against the committed oracle records profiles/r06_parity_cenx_varlen{8,1}.json
  workload_b            (N = 1) the same 50 000 reads with POINT substitutions as copy-specific variants (var_len 1: SURVEY §8(d)'s
                        literal model, simulate_tandem_repeat.py:15-30 — few copy-specific k-mers, E per base collapses): its own
                        ms_per_step, stage split, counters and roofline figures, and every figure of the step (A1 table, rare set,
                        clouds, all pair emissions, every selected edge, the unique k-mers) against the committed oracle record of
                        those reads (profiles/r05_parity_50k_varlen1.json, tools/parity_record.py)

`python bench.py --gpus 8 --reads 62500` is BASELINE configs[3]'s line: 500 000 reads sharded 8 ways.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

K = 19
COVERAGE = 32
VAR_LEN = 8
PARAMS = dict(k=K, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def synth_kwargs(total_reads, seed):
    # M copies of the 2055-bp unit so that aligned bases / (M * 2055) ~= 32 (reads average ~19.5 kb aligned)
    return dict(seed=seed, n_units=max(24, int(round(0.3 * total_reads))), var_len=VAR_LEN)


def cpu_baseline(a, pk, engine):
    """The CPU path (oracle/c/cf_oracle_mt.c, the C restatement of the reference's stage 2) timed on THIS host on the
    benchmark's own reads.  A1-A3 run whole on every core; A5 + A6 (98 % of the CPU time) run for ONE partition of the
    first k-mers (a % n_parts == part — dist_cnt[d][a] is a's own dict in the reference, so a partition is an independent
    1 / n_parts of the same work) and are scaled by n_parts: a bounded sample of the same workload, same reads, same
    emissions per base.  The GPU runs the same partition and must give the same numbers (sample_matches_gpu).
    Second leg: the same partitioned A5 + A6 on ONE thread (distance stage only)."""
    from oracle import cport
    up, us, ue, _ = pk.units(1)
    ncpu = os.cpu_count()
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    n_parts, part = a.cpu_parts, a.cpu_parts // 3
    with cport.Stage2State(pk.bases, pk.read_off, up, us, ue, K, PARAMS["max_nonuniq"], PARAMS["lo"], PARAMS["hi"], threads=0) as st:
        w = st.dist_part(part, n_parts, 0, 2 ** 62, PARAMS["min_d"], PARAMS["max_d"], PARAMS["min_cov"], PARAMS["rel_threshold"], threads=0)
        t_a13 = st.secs_count_select + st.secs_clouds + w["secs_postings"]
        t_all = t_a13 + n_parts * w["secs"]
        # the same partition on the GPU (clouds of the last step are still resident)
        engine.reset_unique()
        ne = engine.dist_edges(0, 2 ** 62, PARAMS["min_d"], PARAMS["max_d"], PARAMS["min_cov"], PARAMS["rel_threshold"], part, n_parts, edge_cap=w["n_edges"] + 16)
        gs = engine.stats()
        match = (ne, gs["n_emissions"], gs["n_unique"], engine.edges_checksum()) == (w["n_edges"], w["n_emissions"], w["n_unique"], w["edge_checksum"]) \
            and (gs["n_kmers"], gs["n_cloud_entries"]) == (st.counters["n_rare"], st.counters["n_cloud_entries"])
        allc = dict(value=pk.n_bases / t_all, unit="bases/s", cores=ncpu, kind="port",
                    sample=f"the benchmark's own {pk.n_reads} reads / {pk.n_bases} bases: A1-A3 whole ({t_a13:.1f} s) + A5/A6 for the first k-mers a % {n_parts} == {part} "
                           f"({w['n_emissions']} pair emissions, {w['secs']:.1f} s) x {n_parts}; oracle/c/cf_oracle_mt.c OpenMP, {ncpu} threads",
                    emissions_per_s=w["n_emissions"] / w["secs"], secs=dict(count_select=st.secs_count_select, clouds=st.secs_clouds,
                                                                           postings=w["secs_postings"], dist_part=w["secs"]),
                    sample_matches_gpu=bool(match),
                    #  the A5/A6 share of `value` is ONE partition's time x n_parts, not a measured whole run
                    extrapolated=True, scale_factor=n_parts, measured_whole_run=measured_cpu_whole_run(pk, a))
        n1 = a.cpu_parts_1t
        w1 = st.dist_part(n1 // 3, n1, 0, 2 ** 62, PARAMS["min_d"], PARAMS["max_d"], PARAMS["min_cov"], PARAMS["rel_threshold"], threads=1)
        one = dict(value=pk.n_bases / (n1 * w1["secs"]), unit="bases/s", cores=1, kind="port",
                   sample=f"the same reads, distance stage only (A5/A6; A1-A3 of 1 Gb on one thread do not fit a bounded sample): first k-mers a % {n1} == {n1 // 3} "
                          f"({w1['n_emissions']} pair emissions, {w1['secs']:.1f} s) x {n1}; oracle/c/cf_oracle_mt.c, 1 thread",
                   emissions_per_s=w1["n_emissions"] / w1["secs"], extrapolated=True, scale_factor=n1)
    out = dict(allc)                 # the headline leg: every core of the host
    out["host_cpus"] = os.cpu_count()
    out["legs"] = [one, allc]
    return out


def cpu_baseline_weak(a):
    """N > 1: the CPU leg on rank 0's host for the per-GPU workload of this weak-scaling line — the N = 1
    configuration's reads (a.reads reads of the array sized for them; the shard of an N-times longer array has 1 / N of the coverage
    and no rare window of its own, so it is no workload by itself).  Same bounded sample as at N = 1: A1-A3 whole + one first-k-mer
    partition x n_parts, every core; the other ranks wait in the closing barrier meanwhile."""
    from centroflye_amd import _host
    from oracle import cport
    pk = _host.synth(n_reads=a.reads, **synth_kwargs(a.reads, a.seed))
    up, us, ue, _ = pk.units(1)
    ncpu = os.cpu_count()
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    n_parts, part = a.cpu_parts, a.cpu_parts // 3
    with cport.Stage2State(pk.bases, pk.read_off, up, us, ue, K, PARAMS["max_nonuniq"], PARAMS["lo"], PARAMS["hi"], threads=0) as st:
        w = st.dist_part(part, n_parts, 0, 2 ** 62, PARAMS["min_d"], PARAMS["max_d"], PARAMS["min_cov"], PARAMS["rel_threshold"], threads=0)
        t_a13 = st.secs_count_select + st.secs_clouds + w["secs_postings"]
        t_all = t_a13 + n_parts * w["secs"]
    return dict(value=pk.n_bases / max(t_all, 1e-9), unit="bases/s", cores=ncpu, kind="port", host_cpus=os.cpu_count(), extrapolated=True, scale_factor=n_parts,
                emissions_per_s=w["n_emissions"] / max(w["secs"], 1e-9),
                sample=f"rank 0's host, the per-GPU workload of this weak-scaling line = the N = 1 configuration ({pk.n_reads} reads / {pk.n_bases} bases): A1-A3 whole "
                       f"({t_a13:.1f} s) + A5/A6 for the first k-mers a % {n_parts} == {part} ({w['n_emissions']} pair emissions, {w['secs']:.1f} s) x {n_parts}; "
                       f"oracle/c/cf_oracle_mt.c OpenMP, {ncpu} threads; compare with value / n_gpus")


def measured_cpu_whole_run(pk, a):
    """The committed 64 / 64-partition CPU run of the same reads (profiles/r03_full_parity.json, tools/full_parity.py): total seconds
    of every partition, when this workload is the one it was taken on — the measured counterpart of the extrapolated leg."""
    path = os.path.join(ROOT, "profiles", "r03_full_parity.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        fp = json.load(f)
    wl = fp.get("workload", {})
    if (wl.get("reads"), wl.get("seed")) != (a.reads, a.seed) or fp.get("n_bases") != pk.n_bases or "cpu_total_secs" not in fp:
        return None
    return dict(secs=fp["cpu_total_secs"], bases_per_s=pk.n_bases / fp["cpu_total_secs"], threads=fp.get("host_cpus"), source="profiles/r03_full_parity.json")


def pmc_traffic(a, world, stored, out):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes — only for the configuration they were taken on."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_dist_kernel.json")))      # the latest round's passes
    path = cands[-1] if cands else ""
    if world != 1 or a.reads != 50000 or a.seed != 2 or stored != out["local_edges"] or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("traffic_bytes_per_launch")


def committed_parity(out, engine, a, world):
    """The result of this run against profiles/r03_full_parity.json: the 64 / 64 CPU run of the oracle on the same reads
    (tools/full_parity.py), i.e. every pair emission and every selected edge of BASELINE configs[2]."""
    path = os.path.join(ROOT, "profiles", "r03_full_parity.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as f:
        fp = json.load(f)
    wl = fp["workload"]
    if (wl["reads"], wl["seed"], wl["params"]) != (a.reads, a.seed, PARAMS) or fp["n_bases"] != out["n_bases"]:
        return None
    got = dict(n_rare=out["n_rare"], n_cloud_entries=out["n_cloud_entries"], n_emissions=out["n_emissions"], n_edges=out["n_edges"],
               n_unique=out["n_unique"], n_distinct=out["n_distinct"], n_kept=out["n_kept"], n_windows=out["n_windows"], n_read_kmers=out["n_read_kmers"])
    if a.edge_cap < 0:
        got["edge_checksum"] = engine.edges_checksum()       # all selected edges are stored: device-side checksum of the rows
    return dict(against="profiles/r03_full_parity.json (oracle/c/cf_oracle_mt.c over all 64 first-k-mer partitions of these reads)",
                checked=sorted(got), match=bool(all(got[k] == fp[k] for k in got)))


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


def timed_steps(sr, steps, warmup, edge_cap):
    """`warmup` untimed steps, then exactly `steps` timed ones between barriers; seconds = max over ranks."""
    E = sr.engine
    outs = []
    for _ in range(warmup):
        outs.append(sr.run(edge_cap=edge_cap, **PARAMS))
    sr.barrier()            # (every library call returns with its stream drained)
    t0 = time.perf_counter()
    kernel_ms = []
    stage_ms = dict(count=0.0, select=0.0, clouds=0.0, postings=0.0, dist=0.0)
    sections = {}
    for _ in range(steps):
        out = sr.run(edge_cap=edge_cap, **PARAMS)
        outs.append(out)
        kernel_ms.append(out["dist_kernel_ms"])
        tm = E.times()
        for k_ in ("count", "select", "clouds", "postings", "dist"):
            stage_ms[k_] += tm[k_ + "_ms"]
        for k_, v in sr.sections.items():
            sections[k_] = sections.get(k_, 0.0) + v
    sr.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = int(sr.allreduce([int(elapsed * 1e9)], "max")[0]) / 1e9       # max over ranks
    return outs, elapsed, kernel_ms, {k: v / max(steps, 1) for k, v in stage_ms.items()}, {k: round(v * 1e3 / max(steps, 1), 3) for k, v in sections.items()}


def pmc_derived():
    """What the committed PMC passes of the dominant kernel say binds it (tools/pmc_summary.py writes `derived` into
    profiles/r0N_pmc_dist_kernel.json from the counters of the bench's own command): the label of roofline.bound and the LDS / L2 /
    HBM-side figures SURVEY §8(d) asks to see next to the HBM fraction."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_dist_kernel.json")))
    if not cands:
        return None, None
    with open(cands[-1]) as f:
        d = json.load(f).get("derived")
    return (d, os.path.relpath(cands[-1], ROOT)) if d else (None, None)


def extra_workload(a, sr, rank, world, kw, steps, what, record):
    """Another read set through the same step: its own ms_per_step, stage split, counters and roofline figures, and EVERY figure of the
    step (A1 table, rare set, clouds, all pair emissions, every selected edge, the unique k-mers) against a committed oracle record of
    those reads (tools/parity_record.py)."""
    from centroflye_amd import _host
    E = sr.engine
    pk = _host.synth(cand_offset=rank, cand_stride=world, **kw)
    sr.load(pk, 1)
    first = sr.run(edge_cap=0, **PARAMS)
    edge_cap = int(first["local_edges"]) + 1024
    outs, elapsed, kernel_ms, stage_ms, sections = timed_steps(sr, steps, 1, edge_cap)
    out = outs[-1]
    stored = min(out["local_edges"], edge_cap)
    ms = elapsed * 1e3 / max(steps, 1)
    mean_k = float(np.mean(kernel_ms)) if kernel_ms else 0.0
    alg = 4 * out["dist_cloud_entries"] + 4 * out["local_emissions"] + 16 * stored
    b_alg = (out["n_bases"] + 16 * out["n_read_kmers"]) + 16 * out["n_distinct"] + (out["n_bases"] + 8 * out["n_windows"] + 4 * out["n_cloud_entries"]) \
        + (4 * out["n_cloud_entries"] + 4 * out["n_emissions"]) + 16 * min(out["n_edges"], edge_cap * world)
    up = np.asarray(pk.units(1)[0])
    res = {"workload": f"{what}: {out['n_bases']} bases in {pk.n_reads} reads per GPU ({float(np.diff(up).mean()):.1f} units per read, {int(np.diff(up).max())} at most), "
                       f"{out['n_rare']} rare k-mers, {out['n_emissions'] / max(out['n_bases'], 1):.1f} pair emissions per base, {out['n_emissions'] / max(out['n_rare'], 1):.0f} per rare k-mer",
           "value": out["n_bases"] * steps / elapsed, "unit": "bases/s", "steps": steps, "ms_per_step": ms,
           "stage_ms_per_step": stage_ms, "host_section_ms_per_step": sections,
           "counters": {k: out[k] for k in ("n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_rare", "n_cloud_entries", "n_emissions", "n_edges", "n_unique", "n_dist_passes")},
           "roofline": {"kernel": "cf_dist_kernel", "kernel_ms": mean_k, "algorithmic_bytes_per_launch": alg, "achieved": alg / (mean_k * 1e-3) / 1e9 if mean_k else 0.0,
                        "unit": "GB/s", "frac": alg / (mean_k * 1e-3) / 1e9 / HBM_PEAK_GBS if mean_k else 0.0,
                        "pair_emissions_per_s": out["local_emissions"] / (mean_k * 1e-3) if mean_k else 0.0,
                        "whole_step_algorithmic_bytes": b_alg, "whole_step_frac": b_alg / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
                        "dist_kernel_share_of_step": mean_k / ms if ms else 0.0},
           "steps_identical": bool(all(all(o[k] == out[k] for k in ("n_edges", "n_emissions", "n_rare", "n_unique", "n_cloud_entries")) for o in outs))}
    path = os.path.join(ROOT, "profiles", record)
    res["parity_vs_committed_oracle"] = None
    if world == 1 and os.path.exists(path):
        with open(path) as f:
            rec = json.load(f)
        wl, part = rec["workload"], rec["partition"]
        same_wl = (wl["reads"], wl["seed"], wl["var_len"], wl["n_units"]) == (kw["n_reads"], kw["seed"], kw["var_len"], kw["n_units"]) \
            and all(kw.get(k_) == v for k_, v in wl.get("synth", {}).items())
        if same_wl and (rec["params"], part["n_parts"]) == (PARAMS, 1) and rec["n_bases"] == out["n_bases"]:
            # the timed steps left the table, the rare set, the clouds, the edges and the unique bitmap of the last step resident
            got = dict(n_windows=out["n_windows"], n_read_kmers=out["n_read_kmers"], n_distinct=out["n_distinct"], n_kept=out["n_kept"], n_rare=out["n_rare"],
                       n_cloud_entries=out["n_cloud_entries"], rare_checksum=E.checksum("kmers")[0], cloud_checksum=E.checksum("clouds")[0])
            gp = dict(n_emissions=out["n_emissions"], n_edges=out["n_edges"], n_unique=out["n_unique"], edge_checksum=E.edges_checksum(),
                      unique_kmers_checksum=E.checksum("unique")[0])
            E.count_kmers(K)          # (select_rare compacts the table it reads: the A1 table of the step is rebuilt for its checksum)
            got["table_checksum"] = E.checksum("table")[0]
            res["parity_vs_committed_oracle"] = dict(
                against=f"profiles/{record} (oracle/c/cf_oracle_mt.c: A1-A3 whole and every first k-mer of these reads)",
                checked=sorted(got) + sorted(gp), match=bool(all(got[k] == rec[k] for k in got) and all(gp[k] == part[k] for k in gp)))
            res["first_kmers"] = part["n_first_kmers"]
            res["dist_passes_per_first_kmer"] = out["n_dist_passes"] / max(part["n_first_kmers"], 1)
    return res


def workload_b(a, sr, rank, world):
    """The bench's reads with point substitutions (var_len 1) through the same step; checked against the committed oracle record."""
    kw = dict(synth_kwargs(a.reads * world, a.seed), var_len=1, n_reads=a.reads)
    return extra_workload(a, sr, rank, world, kw, a.steps_b,
                          f"{a.reads} reads per GPU, the same generator and seed with var_len 1 (point substitutions as copy-specific variants, SURVEY §8(d)'s literal model)",
                          "r05_parity_50k_varlen1.json")


CENX = dict(n_reads=1000, seed=5, n_units=1500, mean_len=100000.0, max_len=1000000)      # (tools/parity_record.py --synth mean_len=100000.0,max_len=1000000)


def workload_c(a, sr, rank, world):
    """BASELINE configs[4]'s SHAPE (the real CHM13 cenX reads are not available to the build): a 1 500-unit array at coverage 32 read by
    ultra-long reads — mean 100 kb, ~47 units per read, up to ~170: every distance up to max_distance = 150 occurs, a first k-mer has
    ~60 000 pair emissions (the default workload: ~20 000, reads of ~10 units).  Reference: README.md:59-75, run_all_cenX.sh:17-22,
    distance_based_kmer_recruitment.py:85-149.  Both copy-specific variant models: var_len 8 and 1."""
    import glob
    out = {}
    for vl in (8, 1):
        out[f"var_len_{vl}"] = extra_workload(a, sr, rank, world, dict(CENX, var_len=vl, n_reads=a.reads_c, n_units=a.units_c), a.steps_c,
                                              f"cenX-shaped reads (1 500-unit array, coverage 32, reads of mean 100 kb), var_len {vl}", f"r06_parity_cenx_varlen{vl}.json")
    # HBM-side bytes of one launch of the dominant kernel on these reads, from the committed PMC passes (tools/profile_round.sh: pmcc_*)
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_dist_kernel_workload_c.json")))
    w8 = out["var_len_8"]
    w8["roofline"]["traffic"] = None
    if cands and (a.reads_c, a.units_c) == (CENX["n_reads"], CENX["n_units"]):
        with open(cands[-1]) as f:
            pm = json.load(f)
        w8["roofline"].update(traffic=pm.get("traffic_bytes_per_launch"), counters_from=os.path.relpath(cands[-1], ROOT),
                              lds=(pm.get("derived") or {}).get("lds"), issue_per_cycle_per_cu=(pm.get("derived") or {}).get("issue_per_cycle_per_cu"),
                              hbm_side_gbps=(pm.get("derived") or {}).get("hbm_side_gbps"))
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50000, help="reads per GPU")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--edge-cap", type=int, default=-1, help="edges stored per GPU (all are counted); -1 = every selected edge")
    ap.add_argument("--cpu-parts", type=int, default=64, help="all-core CPU leg: A5/A6 run for 1 of this many first-k-mer partitions of the bench's reads")
    ap.add_argument("--cpu-parts-1t", type=int, default=1024, help="one-thread CPU leg: the same with this many partitions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--transfer-steps", type=int, default=1, help="extra steps timed with H2D / D2H inside (0 = skip)")
    ap.add_argument("--d2h-edge-bytes", type=int, default=4 << 30, help="stored edges copied back in the transfer-inclusive steps, in bytes")
    ap.add_argument("--place", action="store_true", help="(default at N = 1) also run stage 3 (A4 + A8/A9 placement) once and report it with the end-to-end rate")
    ap.add_argument("--no-place", action="store_true", help="skip stage 3")
    ap.add_argument("--rr", action="store_true", help="also time read recruitment (SURVEY 8(f) rank 4) on the same reads + as many random ones (N = 1)")
    ap.add_argument("--param", action="append", default=[], help="library knob name=value (cf_set_param)")
    ap.add_argument("--force-exchange", action="store_true", help="N = 1 only: run the multi-GPU exchange path (bucketing, all-to-all, all-gathers, gathered view) through a one-rank RCCL communicator, to price it without wire time")
    ap.add_argument("--steps-b", type=int, default=3, help="timed steps of workload_b (the same reads with var_len 1); 0 = skip")
    ap.add_argument("--reads-c", type=int, default=CENX["n_reads"], help="test hook: reads of workload_c"); ap.add_argument("--units-c", type=int, default=CENX["n_units"], help="test hook: array units of workload_c")
    ap.add_argument("--steps-c", type=int, default=5, help="timed steps of workload_c (cenX-shaped reads: BASELINE configs[4]'s regime), N = 1 only; 0 = skip")
    ap.add_argument("--lib", default=None, help="test hook: another build of libcfhip (the CPU suite passes the host-emulated one to check this harness)")
    return ap.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as children (this process never initialises HIP)."""
    idf = os.path.join(tempfile.gettempdir(), f"cfcomm_{os.getuid()}_{os.getpid()}_{int(time.time() * 1e3)}.id")
    nonce = os.urandom(8).hex()      # a fresh token per launch: a stale rendezvous record under a reused name is another launch's
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), CF_COMM_ID_FILE=idf, CF_COMM_NONCE=nonce)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # fail fast: a rank that dies leaves the others waiting in a collective — end them, exit non-zero (this process never
    # touches a GPU, so it may kill its children)
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = max(rc, abs(code))
    if rc:
        for p in live:
            p.terminate()
        t_end = time.time() + 10
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench.py: a rank exited with code {rc}; the other ranks were stopped", file=sys.stderr)
    if os.path.isdir(idf):          # (the emulator's file transport of --lib uses a directory)
        import shutil
        shutil.rmtree(idf, ignore_errors=True)
    elif os.path.exists(idf):
        os.remove(idf)
    sys.exit(rc)


def main():
    a = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        launch_ranks(a)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    from centroflye_amd import _host
    from centroflye_amd.sharded import ShardedRecruiter, default_rendezvous

    total_reads = a.reads * world
    t0 = time.time()
    pk = _host.synth(n_reads=a.reads, cand_offset=rank, cand_stride=world, **synth_kwargs(total_reads, a.seed))
    t_synth = time.time() - t0
    lib = None
    if a.lib:
        from centroflye_amd import _lib
        lib = _lib.load(a.lib)
    sr = ShardedRecruiter(local_rank if not a.lib else 0, lib=lib, rank=rank, world=world, force_exchange=a.force_exchange and world == 1)
    E = sr.engine
    for p in a.param:
        name, val = p.split("=")
        E.set_param(name, int(val))
    t0 = time.time()
    sr.load(pk, 1)          # reads resident in HBM before the timed region
    t_load = time.time() - t0

    # how many edges a step selects on this rank (an untimed pass; everything else about it is a normal step)
    edge_cap = a.edge_cap
    if edge_cap < 0:
        first = sr.run(edge_cap=0, **PARAMS)
        edge_cap = int(first["local_edges"]) + 1024
    outs, elapsed, kernel_ms, stage_ms, sections = timed_steps(sr, a.steps, a.warmup, edge_cap)

    out = outs[-1]
    same = all(all(o[k] == out[k] for k in ("n_edges", "n_emissions", "n_rare", "n_unique", "n_cloud_entries")) for o in outs)
    n_bases = out["n_bases"]
    stored = min(out["local_edges"], edge_cap)

    # the same step with the host hand-over inside the timed region: H2D of the packed reads, D2H of the
    # rare set, the unique mask and stored edges (up to --d2h-edge-bytes of them)
    incl = None
    if a.transfer_steps > 0:
        n_back = int(min(stored, a.d2h_edge_bytes // 16))
        sr.barrier()
        t1 = time.perf_counter()
        h2d_s = 0.0
        for _ in range(a.transfer_steps):
            t_h = time.perf_counter()
            sr.load(pk, 1)
            h2d_s += time.perf_counter() - t_h
            o2 = sr.run(edge_cap=edge_cap, **PARAMS)
            rare = sr.rare
            mask = sr.unique_mask
            edges = E.edges(n_back)
            same = same and o2["n_edges"] == out["n_edges"] and rare.size == out["n_rare"] and int(mask.sum()) == out["n_unique"] and edges.shape[0] == n_back
        sr.barrier()
        el2 = time.perf_counter() - t1
        el2 = int(sr.allreduce([int(el2 * 1e9)], "max")[0]) / 1e9
        incl = dict(value=n_bases * a.transfer_steps / el2, unit="bases/s", steps=a.transfer_steps, ms_per_step=el2 * 1e3 / a.transfer_steps,
                    h2d_bytes=int(pk.n_bases + 8 * (pk.n_reads + 1) + 24 * E.n_units),
                    d2h_bytes=int(8 * out["n_rare"] + out["n_rare"] + 16 * n_back), d2h_edges=n_back, edges_stored=int(stored),
                    h2d_gb_per_s=(pk.n_bases + 8 * (pk.n_reads + 1) + 24 * E.n_units) * a.transfer_steps / h2d_s / 1e9,
                    #  what is NOT in this figure: a complete D2H of every stored edge — the CLI never does it
                    # (it streams the edges by first-k-mer partition into the text writer, or skips them with --no-edges)
                    complete_d2h_of_all_edges_bytes=int(16 * stored),
                    note="the D2H brings back the rare set, the unique mask and d2h_edges edge rows; a complete copy of all stored edges would be "
                         "complete_d2h_of_all_edges_bytes (about one second at the 55 GB/s this path reaches), which scripts/distance_based_kmer_recruitment.py "
                         "never needs at once: it writes the edge file partition by partition")

    if rank == 0:
        ms_per_step = elapsed * 1e3 / max(a.steps, 1)
        mean_k_ms = float(np.mean(kernel_ms)) if kernel_ms else 0.0
        # algorithmic bytes of ONE launch of the dominant kernel on this rank: every cloud entry it
        # works on staged once, one 4-byte partner index per pair emission, 16 bytes per stored edge
        alg_bytes = 4 * out["dist_cloud_entries"] + 4 * out["local_emissions"] + 16 * stored
        achieved = alg_bytes / (mean_k_ms * 1e-3) / 1e9 if mean_k_ms > 0 else 0.0
        # B_alg of the whole step, all ranks: [N_b + 16 N_rk] + [16 K_dist] + [N_b + 8 N_w + 4 N_ce] + [4 N_ce + 4 E] + [16 edges stored]
        b_alg = (n_bases + 16 * out["n_read_kmers"]) + 16 * out["n_distinct"] + (n_bases + 8 * out["n_windows"] + 4 * out["n_cloud_entries"]) \
            + (4 * out["n_cloud_entries"] + 4 * out["n_emissions"]) + 16 * min(out["n_edges"], edge_cap * world)
        derived, derived_from = pmc_derived()
        res = {
            "metric": "long-read bases/sec through rare-k-mer recruit+distance",
            "value": n_bases * a.steps / elapsed,
            "unit": "bases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 bases -> u64 k-mers / u32 indices (integer only)",
            "data": "synthetic",
            "config": {"workload": f"{a.reads} synthetic DXZ1-HOR ONT-like reads per GPU ({n_bases} aligned bases in all), "
                                   f"2055-bp unit x {synth_kwargs(total_reads, a.seed)['n_units']} copies, copy-specific variants of var_len {VAR_LEN}, k={K}, "
                                   f"coverage {COVERAGE}, max_distance {PARAMS['max_d']}: count + rare filter + clouds + distance/filter, "
                                   f"{stored} of {out['local_edges']} selected edges stored per GPU (BASELINE configs[1]+[2], stage 2)",
                       "reads_per_gpu": a.reads, "k": K, "var_len": VAR_LEN, "edges_stored": int(stored), "edges_selected": int(out["local_edges"]),
                       "parallelism": f"reads sharded x{world}, first k-mers partitioned x{world}"},
            "value_incl_transfers": incl,
            # `frac` prices the kernel's ALGORITHMIC bytes against the HBM peak (the contract's definition); `bound` names what the
            # counters of the committed PMC passes say limits it — for this kernel LDS round-trip latency at 4 waves per SIMD, not HBM
            # ("the roofline label is formal; say what binds"): `lds`, `l2_hit`, `hbm_side_gbps` are those counters
            "roofline": {"bound": (derived or {}).get("bound", "hbm"), "bound_by_contract": "hbm", "kernel": "cf_dist_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(a, world, stored, out),
                         "lds": (derived or {}).get("lds"), "l2_hit": (derived or {}).get("l2_hit"), "hbm_side_gbps": (derived or {}).get("hbm_side_gbps"),
                         "issue_per_cycle_per_cu": (derived or {}).get("issue_per_cycle_per_cu"), "counters_from": derived_from,
                         "traffic_note": "HBM-side bytes of ONE launch from the committed rocprofv3 --pmc passes of this configuration (profiles/r0N_pmc_dist_kernel.json, the latest round's: "
                                         "(2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate passes, tools/profile_round.sh); counters cannot be read inside a timed run; null when the workload differs",
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": mean_k_ms,
                         "pair_emissions_per_s": out["local_emissions"] / (mean_k_ms * 1e-3) if mean_k_ms else 0.0,
                         "whole_step_algorithmic_bytes": b_alg,
                         "whole_step_frac": b_alg / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)},
            "counters": {k: out[k] for k in ("n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_rare", "n_cloud_entries",
                                             "n_emissions", "n_edges", "n_unique", "n_dist_passes")},
            "stage_ms_per_step": stage_ms,
            "host_section_ms_per_step": sections,
            "exchange_bytes_per_step": int(sr.exchange_bytes),
            "setup_s": {"synth": round(t_synth, 2), "load_h2d": round(t_load, 3)},
            "steps_identical": bool(same),
            "device": E.device_info()["name"].strip(),
        }
        res["parity_vs_committed_oracle"] = committed_parity(out, E, a, world)
        if world == 1 and not a.no_place:
            # BASELINE configs[2]: cloud_contig extension on the same reads with the k-mers selected above
            t1 = time.perf_counter()
            gk = sr.rare[sr.unique_mask]
            E.set_kmers(gk, K)
            E.build_clouds()
            E.filter_clouds(2)
            cls = pk.classify(50000)
            idr = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
            t2 = time.perf_counter()
            rd, pos, s0, s1 = E.place_reads(cls, idr, 2, 2, 10, 3)
            t3 = time.perf_counter()
            res["placement"] = {"reads": int(pk.n_reads), "placed": int((pos >= 0).sum()), "none": int((pos < 0).sum()),
                                "classes": np.bincount(cls, minlength=3).tolist(), "clouds_filter_s": t2 - t1, "place_s": t3 - t2,
                                "place_device_ms": E.times()["place_ms"],
                                "end_to_end_bases_per_s": n_bases / (ms_per_step * 1e-3 + (t3 - t1))}
            res["end_to_end"] = {"value": res["placement"]["end_to_end_bases_per_s"], "unit": "bases/s",
                                 "what": "one stage-2 step (ms_per_step) + A3/A4 on the selected k-mers + greedy placement of all reads (BASELINE configs[2]), reads resident"}
        if world == 1 and a.rr:
            res["read_recruitment"] = rr_leg(E, pk, a.no_cpu_baseline)
        if world == 1 and not a.no_cpu_baseline:
            sr.load(pk, 1)       # (stage 3 installed the placer's k-mer set and clouds)
            sr.run(edge_cap=0, **PARAMS)
            res["cpu_baseline"] = cpu_baseline(a, pk, E)
        else:
            res["cpu_baseline"] = None
    wb = workload_b(a, sr, rank, world) if a.steps_b > 0 else None      # (every rank: its steps hold collectives)
    wc = workload_c(a, sr, rank, world) if (a.steps_c > 0 and world == 1) else None
    if rank == 0:
        res["workload_b"] = wb
        res["workload_c"] = wc
        if world > 1 and not a.no_cpu_baseline:      # (behind the last collective of the other ranks: they wait in the closing barrier)
            res["cpu_baseline"] = cpu_baseline_weak(a)
        print(json.dumps(res), flush=True)
    sr.barrier()
    sr.close()


if __name__ == "__main__":
    main()
