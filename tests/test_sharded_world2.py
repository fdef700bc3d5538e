"""The N > 1 path (reads sharded, device-side owner bucketing, all-to-all count merge, all-gathers, first-k-mer
partition, mask all-reduce) on world_size 2: two processes running the host-emulated kernels, exchanging through the
emulator's file transport behind the C entry points of include/cfhip.h (cf_comm_init, cf_exchange_table,
cf_allgather_kmers, cf_allgather_clouds, cf_allreduce_unique); compared with the C oracle run on the union of the shards."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,extra", [(2, []), (3, []), (2, ["4096"]), (3, ["1008", "1"])])
def test_sharded_equals_single_process_oracle(emu_lib, tmp_path, world, extra):
    """extra = [comm_round_bytes [, comm_self_p2p]]: every pair's messages cut into many small rounds (ranks then run
    DIFFERENT numbers of rounds per exchange: the rounds of a pair follow from the pair's byte count alone), and a rank's
    message to itself sent through the transport as well."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), str(r), str(world), str(rdv)] + extra,
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-3000:] + se[-3000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("SHARDED_RESULT ")]
    assert line, outs[0][0][-2000:]
    res = json.loads(line[0].split(" ", 1)[1])
    assert res["world"] == world and res["n_rare"] > 100 and res["n_edges"] > 100 and res["exchange_bytes"] > 0
    assert res["rare"] and res["unique"] and res["counters"] and res["edge_checksum"] and res["steps_identical"], res


@pytest.mark.parametrize("mode,refused", [("exotic_block", True), ("exotic_pass", False), ("exotic_lower", False)])
def test_exotic_windows_are_judged_on_counts_added_over_the_shards(emu_lib, tmp_path, mode, refused):
    """Windows with a symbol other than upper-case A, C, G, T have no 2-bit code; the reference counts them as k-mers of their own
    (distance_based_kmer_recruitment.py:47-53).  A shard alone cannot tell whether one is rare: presence and multiplicity add over
    the shards, and every rank takes the same decision as the single-GPU entry point would on the whole read set (round-3
    advice: a lone N on one shard used to stop the multi-GPU run)."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), str(r), "2", str(rdv), str(1 << 28), "0", mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-3000:] + se[-3000:]
        res = json.loads([ln for ln in so.splitlines() if ln.startswith("EXOTIC_RESULT ")][0].split(" ", 1)[1])
        assert res["refused"] == refused, res


def test_rare_kmers_with_an_N_travel_through_the_sharded_path(emu_lib, tmp_path):
    """Fixture "exotic_rare" dealt over two ranks: the windows with an N are rare only on the ADDED counts; their text is exchanged,
    every rank installs the same pseudo-codes behind the gathered rare list and adds its own units' cloud entries — rare set, edges
    (checksum over all ranks), unique k-mers and counters equal the single-process run on the whole report."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), str(r), "2", str(rdv), str(1 << 28), "0", "exotic_real"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=1500))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-3000:] + se[-3000:]
    res = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("EXOTIC_REAL ")][0].split(" ", 1)[1])
    assert res["n_extra"] >= 30 and res["edges_with_such_a_kmer"] > 0
    assert all(res[k] is True for k in ("same_strings", "same_codes", "n_rare", "n_edges", "edge_checksum", "unique", "n_emissions", "n_cloud_entries")), res
