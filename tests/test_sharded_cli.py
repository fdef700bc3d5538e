"""The sharded path behind the reference's command line: `CF_GPUS=N python
scripts/distance_based_kmer_recruitment.py ...` starts N ranks (centroflye_amd/sharded_cli.py) that shard the parsed report,
run A1-A6 with the exchanges of centroflye_amd/sharded.py and write the reference's two files
(distance_based_kmer_recruitment.py:152-171).  CPU: two / three ranks on the host-emulated kernels with the file transport —
k-mer file byte for byte and the edge lines as a set against the REFERENCE's goldens (fixture exotic_rare, whose goldens were
taken with --max-distance 2, a size the emulator runs) and against the numpy oracle (lowcov); GPU (-m gpu): one rank through
the whole exchange path (RCCL communicator of one rank) on every fixture against the reference's goldens."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import canon
import fixtures
from centroflye_amd import _host, sharded_cli
from oracle import recruit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = [sys.executable, os.path.join(ROOT, "tests", "sharded_cli_worker.py")]


def _argv(report, outdir, p2, extra=()):
    return ["--ncrf", report, "--coverage", str(p2["coverage"]), "--min-coverage", str(p2["min_coverage"]), "--outdir", outdir, "-k", str(p2["k"]),
            "--max-distance", str(p2["max_distance"]), "--min-distance", str(p2["min_distance"])] + list(extra)


def _launch(argv, n, monkeypatch, sub_edges=0):
    monkeypatch.setenv("OMP_NUM_THREADS", "1")
    monkeypatch.setenv("CF_TEST_SUB_EDGES", str(sub_edges))
    monkeypatch.delenv("CF_PACK_CACHE", raising=False)
    return sharded_cli.launch(argv, n, rank_cmd=WORKER)


def test_shards_are_contiguous_runs_of_reads_with_equal_bases(report):
    pk = _host.parse_report(report("lowcov"), keep_rows=False)
    for world in (1, 2, 3, 7, 64):
        cuts = sharded_cli.shard_bounds(pk.read_off, world)
        assert cuts[0] == 0 and cuts[-1] == pk.n_reads and all(a <= b for a, b in zip(cuts, cuts[1:]))
        shards = [sharded_cli.ReadShard(pk, cuts[r], cuts[r + 1]) for r in range(world)]
        assert np.array_equal(np.concatenate([s.bases for s in shards]), pk.bases)
        assert sum(s.n_reads for s in shards) == pk.n_reads
        up, us, ue, _ = pk.units(1)
        assert sum(s.units(1)[1].size for s in shards) == us.size
        for s in shards:
            sup, sus, sue, _ = s.units(1)
            assert sup[0] == 0 and sup[-1] == sus.size and s.read_off[0] == 0 and s.read_off[-1] == s.n_bases
            if sus.size:
                assert sus.min() >= 0 and sue.max() <= s.n_bases
                u0 = int(up[s.lo])
                assert np.array_equal(s.bases[sus[0]:sue[0]], pk.bases[us[u0]:ue[u0]])
        if world <= 3:
            sizes = [s.n_bases for s in shards]
            assert max(sizes) - min(sizes) < 3 * int(np.diff(pk.read_off).max())


@pytest.mark.parametrize("world,sub_edges", [(2, 0), (3, 700)])
def test_cf_gpus_ranks_write_the_oracles_files_on_emulated_kernels(emu_lib, report, oracle_stage2, tmp_path, monkeypatch, world, sub_edges):
    """lowcov with --max-distance 2; sub_edges = 700: every rank streams its first k-mers in several sub-partitions (a partition that
    does not fit is started over with more of them)."""
    name = "lowcov"
    p2 = fixtures.stage2_params(name)
    p2["max_distance"] = 2
    assert _launch(_argv(report(name), str(tmp_path), p2, ["--metrics"]), world, monkeypatch, sub_edges) == 0
    records, alns, lens, res, _ = oracle_stage2(name, max_distance=2)
    with open(tmp_path / f"unique_kmers_min_edge_cov_{p2['min_coverage']}.txt") as f:
        assert f.read() == recruit.kmers_file_text(res["rare"], res["unique"], p2["k"])
    with open(tmp_path / f"unique_edges_min_edge_cov_{p2['min_coverage']}.txt") as f:
        lines = f.read().splitlines()
    assert sorted(lines) == recruit.edges_file_lines(res["rare"], res["edges"], p2["k"]) and len(lines) > 2 * max(sub_edges, 300)
    with open(tmp_path / "stage2_metrics.json") as f:
        m = json.load(f)
    assert m["world"] == world and sum(m["shard_reads"]) == len(records) and m["pack_from_cache"] and m["exchange_bytes"] > 0
    assert m["stats"]["n_edges"] == len(lines) and m["stats"]["n_unique"] == int(np.count_nonzero(res["unique"]) if res["unique"].dtype == bool else len(res["unique"]))
    assert not [p for p in os.listdir(tmp_path) if ".tmp" in p or p.startswith(".cfpack")]


def test_a_chunk_smaller_than_one_first_kmers_edges(emu_lib, report, oracle_stage2, tmp_path, monkeypatch):
    """sub_edges = 1: fewer rows than ONE first k-mer has edges — the sub-partitions stop at one first k-mer each and the chunk grows instead
    (round 5, tools/fuzz_cli_vs_reference.py: their number grew until world * n_sub left the 32-bit range: 'bad partition').  A rare window of
    exactly 9 reads keeps the k-mer set (and the number of launches on the emulator) small."""
    name = "lowcov"
    over = dict(max_distance=2, bottom=1.0, top=1.0, coverage=9, kmer_survival_rate=1.0, min_coverage=1)
    p2 = dict(fixtures.stage2_params(name), **over)
    extra = ["--bottom", "1.0", "--top", "1.0", "--kmer-survival-rate", "1.0"]
    assert _launch(_argv(report(name), str(tmp_path), p2, extra), 2, monkeypatch, 1) == 0
    records, alns, lens, res, _ = oracle_stage2(name, **over)
    assert 10 < res["rare"].size < 2000 and res["edges"].shape[0] > 50
    with open(tmp_path / "unique_kmers_min_edge_cov_1.txt") as f:
        assert f.read() == recruit.kmers_file_text(res["rare"], res["unique"], p2["k"])
    with open(tmp_path / "unique_edges_min_edge_cov_1.txt") as f:
        assert sorted(f.read().splitlines()) == recruit.edges_file_lines(res["rare"], res["edges"], p2["k"])


def test_more_ranks_than_reads(emu_lib, report, tmp_path, monkeypatch):
    """Two reads, three ranks: one rank has no read at all (an empty shard goes through the table exchange, the gathers and the distance stage
    like any other); the files are the numpy oracle's for those two records."""
    from oracle import ncrf
    with open(report("lowcov")) as f:
        lines = f.read().split("\n")
    recs = [i for i, ln in enumerate(lines) if ln and not ln.startswith("#")][:4]
    small = tmp_path / "two_reads.ncrf"
    small.write_text("\n".join(lines[i] for i in recs) + "\n")
    p2 = dict(fixtures.STAGE2_DEFAULTS, coverage=2, min_coverage=2, max_distance=1)      # the window 0.9 .. 3 x 2 x 0.34 keeps k-mers of one or two reads
    out = tmp_path / "out"
    assert _launch(_argv(str(small), str(out), p2), 3, monkeypatch) == 0
    records, _, _ = ncrf.parse_report(str(small))
    assert len(records) == 2
    res = recruit.stage2(records, k=p2["k"], bottom=p2["bottom"], top=p2["top"], coverage=p2["coverage"], kmer_survival_rate=p2["kmer_survival_rate"],
                         max_nonuniq=p2["max_nonuniq"], min_n=p2["min_nreads"], max_n=p2["max_nreads"], min_d=p2["min_distance"], max_d=p2["max_distance"],
                         min_coverage=p2["min_coverage"])
    with open(out / "unique_kmers_min_edge_cov_2.txt") as f:
        assert f.read() == recruit.kmers_file_text(res["rare"], res["unique"], p2["k"])
    with open(out / "unique_edges_min_edge_cov_2.txt") as f:
        got = f.read().splitlines()
    assert sorted(got) == recruit.edges_file_lines(res["rare"], res["edges"], p2["k"]) and len(got) > 100


def test_cf_gpus_2_reproduces_the_reference_files_with_rare_N_kmers_on_emulated_kernels(emu_lib, fx_dir, tmp_path, monkeypatch):
    """Fixture exotic_rare (goldens written by the reference itself): 35 of the unique k-mers hold an N; they travel through the
    sharded path as strings and reach both files."""
    with open(os.path.join(ROOT, "tests", "golden", "exotic_rare.json")) as f:
        g = json.load(f)
    rep = fixtures.make_report("exotic_rare", fx_dir)
    assert fixtures.sha256_file(rep) == g["report_sha256"]
    p2 = g["stage2"]
    assert _launch(_argv(rep, str(tmp_path), p2), 2, monkeypatch) == 0
    with open(tmp_path / f"unique_kmers_min_edge_cov_{p2['min_coverage']}.txt", "rb") as f:
        data = f.read()
    assert hashlib.sha256(data).hexdigest() == g["unique_kmers"]["sha256"] and data.count(b"N") >= 30
    with open(tmp_path / f"unique_edges_min_edge_cov_{p2['min_coverage']}.txt") as f:
        elines = f.read().splitlines()
    assert len(elines) == g["edges"]["n"] and canon.edge_lines_digest(elines) == g["edges"]["digest"]


def test_ranks_of_a_launcher_write_the_same_files(emu_lib, report, oracle_stage2, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 <stage script> ...`: the ranks find RANK / WORLD_SIZE / LOCAL_RANK in their
    environment (no CF_GPUS parent, no pack cache: every rank parses the report itself) and derive the rendezvous from their common
    parent; torch is only the launcher."""
    pytest.importorskip("torch")
    import socket
    name = "lowcov"
    p2 = fixtures.stage2_params(name)
    p2["max_distance"] = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           WORKER[1]] + _argv(report(name), str(tmp_path), p2)
    env = {k: v for k, v in os.environ.items() if k not in ("CF_PACK_CACHE", "CF_COMM_ID_FILE", "CF_COMM_NONCE", "CF_GPUS")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    records, alns, lens, res, _ = oracle_stage2(name, max_distance=2)
    with open(tmp_path / f"unique_kmers_min_edge_cov_{p2['min_coverage']}.txt") as f:
        assert f.read() == recruit.kmers_file_text(res["rare"], res["unique"], p2["k"])
    with open(tmp_path / f"unique_edges_min_edge_cov_{p2['min_coverage']}.txt") as f:
        assert sorted(f.read().splitlines()) == recruit.edges_file_lines(res["rare"], res["edges"], p2["k"])


def test_cf_gpus_no_edges_and_a_dead_rank(emu_lib, report, tmp_path, monkeypatch):
    name = "lowcov"
    p2 = fixtures.stage2_params(name)
    p2["max_distance"] = 2
    out = tmp_path / "ok"
    assert _launch(_argv(report(name), str(out), p2, ["--no-edges"]), 2, monkeypatch) == 0
    assert sorted(os.listdir(out)) == [f"unique_kmers_min_edge_cov_{p2['min_coverage']}.txt"]
    # a rank that dies (here: every rank, on a report that does not exist for them) ends the launch with a non-zero code
    monkeypatch.setenv("CF_TEST_SUB_EDGES", "0")
    bad = sharded_cli.launch(_argv(report(name), str(tmp_path / "bad"), p2), 2, rank_cmd=[sys.executable, "-c", "import sys; sys.exit(3)"])
    assert bad == 3


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(fixtures.FIXTURES) + ["exotic_rare"])
def test_sharded_stage_script_reproduces_the_reference_files_on_the_gpu(name, fx_dir, tmp_path):
    """`CF_SHARDED=1 python -u scripts/distance_based_kmer_recruitment.py ...`: this process is rank 0 of 1 and goes through the whole
    exchange path (a one-rank RCCL communicator, the message to itself through ncclSend / ncclRecv); `CF_GPUS=1` must be the plain path."""
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.json")) as f:
        g = json.load(f)
    rep = fixtures.make_report(name, fx_dir)
    p2 = dict(fixtures.STAGE2_DEFAULTS)
    p2.update(g["stage2"] if "stage2" in g else fixtures.stage2_params(name))
    out = str(tmp_path / "sharded")
    env = dict(os.environ, CF_SHARDED="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    subprocess.check_call([sys.executable, "-u", os.path.join(ROOT, "scripts", "distance_based_kmer_recruitment.py")] + _argv(rep, out, p2, ["--metrics"]),
                          stdout=subprocess.DEVNULL, env=env)
    with open(os.path.join(out, f"unique_kmers_min_edge_cov_{p2['min_coverage']}.txt"), "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == g["unique_kmers"]["sha256"]
    with open(os.path.join(out, f"unique_edges_min_edge_cov_{p2['min_coverage']}.txt")) as f:
        elines = f.read().splitlines()
    assert len(elines) == g["edges"]["n"] and canon.edge_lines_digest(elines) == g["edges"]["digest"]
    with open(os.path.join(out, "stage2_metrics.json")) as f:
        m = json.load(f)
    assert m["world"] == 1 and m["exchange_bytes"] >= 0 and "table_exchange" in m["sections_s"] and "cloud_gather" in m["sections_s"]


def test_eight_ranks_on_emulated_kernels_some_without_reads(emu_lib, report, oracle_stage2, tmp_path, monkeypatch):
    """World 8 before an 8-GPU node runs it: CF_GPUS=8 on the emulated kernels with the file transport; the fixture has
    fewer records than 8 x 2, so shards are a read or two and every exchange (table all-to-all, gathers, mask reduction) has 8 ends."""
    name = "lowcov"
    p2 = fixtures.stage2_params(name)
    p2["max_distance"] = 2
    assert _launch(_argv(report(name), str(tmp_path), p2, ["--metrics"]), 8, monkeypatch) == 0
    records, alns, lens, res, _ = oracle_stage2(name, max_distance=2)
    with open(tmp_path / f"unique_kmers_min_edge_cov_{p2['min_coverage']}.txt") as f:
        assert f.read() == recruit.kmers_file_text(res["rare"], res["unique"], p2["k"])
    with open(tmp_path / f"unique_edges_min_edge_cov_{p2['min_coverage']}.txt") as f:
        assert sorted(f.read().splitlines()) == recruit.edges_file_lines(res["rare"], res["edges"], p2["k"])
    with open(tmp_path / "stage2_metrics.json") as f:
        m = json.load(f)
    assert m["world"] == 8 and len(m["shard_reads"]) == 8 and sum(m["shard_reads"]) == len(records)
    assert not [p for p in os.listdir(tmp_path) if ".tmp" in p or p.startswith(".cfpack")]


def test_default_rank_command_resolves_the_package_from_a_foreign_directory(report, tmp_path):
    """ `cd elsewhere && CF_GPUS=2 python <repo>/scripts/distance_based_kmer_recruitment.py ...` — how centroFlye.py:172-188
    calls it — starts its ranks with the script's own path: they must get as far as the device (which this container does not have),
    not die on `import centroflye_amd`; and the parent leaves neither its private pack cache nor part files behind."""
    p2 = fixtures.stage2_params("lowcov")
    out = tmp_path / "out"
    elsewhere = tmp_path / "elsewhere"
    elsewhere.mkdir()
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "CF_PACK_CACHE", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(CF_GPUS="2", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "distance_based_kmer_recruitment.py")] + _argv(report("lowcov"), str(out), p2),
                       capture_output=True, text=True, timeout=300, cwd=elsewhere, env=env)
    assert "ModuleNotFoundError" not in p.stderr and "No module named" not in p.stderr, p.stderr[-2000:]
    import torch                                    # (only to ask whether a GPU is here; the package itself never imports it)
    if not torch.cuda.is_available():
        assert p.returncode != 0 and ("cf_create" in p.stderr or "device" in p.stderr.lower() or "hip" in p.stderr.lower()), p.stderr[-2000:]
    assert not [x for x in os.listdir(out) if x.startswith(".cfpack") or ".tmp" in x]


def test_an_unreadable_report_leaves_nothing_behind(tmp_path, monkeypatch):
    """ launch() cleans up on every path out: the private pack cache is removed when the parse raises."""
    monkeypatch.delenv("CF_PACK_CACHE", raising=False)
    out = tmp_path / "out"
    with pytest.raises(Exception):
        sharded_cli.launch(["--ncrf", str(tmp_path / "missing.ncrf"), "--coverage", "32", "--outdir", str(out)], 2, rank_cmd=WORKER)
    assert os.path.isdir(out) and not os.listdir(out)
