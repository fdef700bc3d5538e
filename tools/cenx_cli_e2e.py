#!/usr/bin/env python3
"""GPU-box tool: BOTH drop-in command lines and the polisher export END TO END on a report of BASELINE configs[4]'s SHAPE — a 1 500-unit
array at coverage 32 read by ultra-long reads (mean 100 kb: ~47 units per read, up to ~170; README.md:59-75 and run_all_cenX.sh:17-22 of the
reference: the cenX run that took its Python scripts "9 h") — every output against the CPU oracles, wall times into the record:

  scripts/distance_based_kmer_recruitment.py --ncrf R --coverage 32 --outdir S2 --no-edges     (centroFlye.py:172-188)
      unique_kmers_min_edge_cov_4.txt: count and checksum of the k-mers = the committed record of oracle/c/cf_oracle_mt.c over EVERY first
      k-mer of these reads (profiles/r06_parity_cenx_varlen{8,1}.json, tools/parity_record.py);
  scripts/read_placer.py --ncrf R --genomic-kmers S2/unique_kmers... --outdir S3                (centroFlye.py:196-208)
      read_positions.csv: placed lines in order and the None lines as a set = the C placer (oracle/c/cf_oracle_place.c) on the filtered
      clouds of the same k-mers;
  scripts/eltr_polisher.py --read-placement S3/read_positions.csv --unit U --ncrf R --outdir P --export-only   (centroFlye.py:211-222)
      SHA-256 of every pos_*/read_units.fasta and median_read_unit.fasta = oracle/polisher.py on the oracle's own parse of the report.

  python3 tools/cenx_cli_e2e.py --var-len 8 --out gpurun_out/r06_cenx_cli_e2e_varlen8.json
(The edge file is skipped: 1.8e9 edges are 87 GB of text that nothing reads; the edges themselves are checked by the -m gpu test on the
same record.)"""
import argparse, hashlib, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--var-len", type=int, default=8); ap.add_argument("--out", required=True); ap.add_argument("--keep", action="store_true")
a = ap.parse_args()
with open(os.path.join(ROOT, "profiles", f"r06_parity_cenx_varlen{a.var_len}.json")) as f:
    rec = json.load(f)
wl = rec["workload"]
from centroflye_amd import _host
from oracle import cport, ncrf, polisher
W = tempfile.mkdtemp(prefix="cf_cenx_e2e_")
res = dict(what=__doc__.split("\n")[0], workload=wl, wall_s={}, checks={})
try:
    R = os.path.join(W, "report.ncrf")
    t0 = time.time()
    _host.synth(report_path=R, pack=False, n_reads=wl["reads"], seed=wl["seed"], n_units=wl["n_units"], var_len=wl["var_len"], **wl.get("synth", {}))
    res["wall_s"]["write_report"] = round(time.time() - t0, 2); res["report_bytes"] = os.path.getsize(R)

    def run(name, argv):
        t0 = time.time()
        p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "scripts", argv[0])] + argv[1:], capture_output=True, text=True, cwd=W)
        res["wall_s"][name] = round(time.time() - t0, 2)
        if p.returncode != 0:
            raise SystemExit(f"{name} failed (rc {p.returncode}):\n{p.stderr[-3000:]}")

    S2, S3, PO = os.path.join(W, "s2"), os.path.join(W, "s3"), os.path.join(W, "polish")
    run("distance_based_kmer_recruitment", ["distance_based_kmer_recruitment.py", "--ncrf", R, "--coverage", "32", "--outdir", S2, "--no-edges", "--metrics"])
    kfile = os.path.join(S2, "unique_kmers_min_edge_cov_4.txt")
    gk = _host.read_kmers(kfile, 19)
    with open(os.path.join(S2, "stage2_metrics.json")) as f:
        res["stage2_metrics"] = json.load(f)
    res["n_unique_kmers"] = int(gk.size)
    res["checks"]["unique_kmers_count_and_checksum_vs_oracle_record"] = bool(gk.size == rec["partition"]["n_unique"] and cport.rare_checksum(gk) == rec["partition"]["unique_kmers_checksum"])

    run("read_placer", ["read_placer.py", "--ncrf", R, "--genomic-kmers", kfile, "--outdir", S3])
    csv = os.path.join(S3, "read_positions.csv")
    with open(csv) as f:
        got_lines = f.read().splitlines()
    # the C placer on the filtered clouds of the same k-mers (A3 / A4 of the device: pinned by the parity tests of those rows)
    from conftest import lines_from_placement
    from centroflye_amd.engine import Engine
    t0 = time.time()
    pk = _host.parse_report(R)
    with Engine(0) as e:
        e.load(pk, 1); e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(2)
        cp, ent = e.clouds()
    up = pk.units(1)[0]
    cls = pk.classify(50000)
    rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
    want = cport.place_reads(cls, rank, up, cp, ent, gk.size, 2, 2, 10, 3)
    want_lines = lines_from_placement(pk.ids, *[x.tolist() for x in want])
    res["wall_s"]["c_placer_oracle"] = round(time.time() - t0, 2)
    placed = lambda ls: [x for x in ls if not x.endswith(" None")]
    res["placed"] = len(placed(got_lines)); res["none"] = len(got_lines) - res["placed"]
    res["checks"]["read_positions_vs_c_placer"] = bool(placed(got_lines) == placed(want_lines) and sorted(got_lines) == sorted(want_lines) and res["placed"] > 0.9 * len(got_lines))

    unit = os.path.join(W, "unit.fasta")
    with open(unit, "w") as f:
        f.write(">unit\n" + pk.motifs[0] + "\n")
    run("eltr_polisher_export", ["eltr_polisher.py", "--read-placement", csv, "--unit", unit, "--ncrf", R, "--outdir", PO, "--export-only"])
    t0 = time.time()
    records, _, _ = ncrf.parse_report(R)
    files = polisher.export(records, polisher.read_reported_positions(csv))
    bad = 0
    for p, (units_txt, med_txt) in files.items():
        for name, txt in (("read_units.fasta", units_txt), ("median_read_unit.fasta", med_txt)):
            fn = os.path.join(PO, f"pos_{p}", name)
            if not os.path.exists(fn) or hashlib.sha256(open(fn, "rb").read()).hexdigest() != hashlib.sha256(txt.encode()).hexdigest():
                bad += 1
    n_dirs = sum(1 for d in os.listdir(PO) if d.startswith("pos_"))
    res["wall_s"]["polisher_oracle"] = round(time.time() - t0, 2)
    res["positions_exported"] = n_dirs
    res["checks"]["exported_fasta_vs_oracle"] = bool(bad == 0 and n_dirs == len(files) and n_dirs > 0)
    res["n_bases"] = int(pk.n_bases)
    res["wall_s"]["three_command_lines"] = round(sum(res["wall_s"][k] for k in ("distance_based_kmer_recruitment", "read_placer", "eltr_polisher_export")), 2)
    res["identical"] = all(res["checks"].values())
finally:
    if not a.keep:
        shutil.rmtree(W, ignore_errors=True)
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
with open(a.out, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
sys.exit(0 if res.get("identical") else 1)
