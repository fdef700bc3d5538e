"""Worker of tests/test_sharded_world2.py: one rank of the sharded recruit+distance path on the host-emulated kernels,
talking to its peer through the emulator's file transport (tests/emu/cf_comm_emu.cpp) behind the same C entry points
the RCCL build exports.  Rank 0 checks the result against the C oracle run on the union of all shards.

    python sharded_worker.py RANK WORLD RENDEZVOUS_DIR [COMM_ROUND_BYTES [SELF_P2P]]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

from centroflye_amd import _host, _lib
from centroflye_amd.sharded import ShardedRecruiter
from oracle import cport

SYNTH = dict(seed=17, unit_len=200, monomer_len=50, n_units=60, flank=60000, mean_len=6500, sigma=0.2, min_len=6000,
             max_len=7000, unit_div=0.03, n_prefix=2, n_suffix=2, prefix_threshold=50000)
PARAMS = dict(k=19, max_nonuniq=3, lo=4, hi=14, min_d=1, max_d=2, min_cov=2, rel_threshold=0.8)
READS_PER_RANK = 12


def exotic_real(rank, world, rdv, lib):
    """Fixture "exotic_rare" (k-mers with an N that ARE rare) dealt record by record to the ranks: the sharded run must carry those
    k-mers exactly as the single-process path does on the whole report (same rare set, same edges, same unique k-mers)."""
    import fixtures
    from centroflye_amd import kmers as km
    from centroflye_amd.engine import Engine
    from centroflye_amd.read_kmer_cloud import add_exotic_entries
    full = fixtures.make_report("exotic_rare", rdv)      # (both ranks write the same bytes through tmp + rename)
    with open(full) as f:
        lines = f.read().split("\n")
    recs = [i for i, ln in enumerate(lines) if ln and not ln.startswith("#")]
    mine = set()
    for j in range(0, len(recs), 2):
        if (j // 2) % world == rank:
            mine.update(recs[j:j + 2])
    part = os.path.join(rdv, f"part{rank}.ncrf")
    with open(part, "w") as f:
        f.write("\n".join(ln for i, ln in enumerate(lines) if i in mine or ln.startswith("#") or not ln))
    P = dict(PARAMS, lo=5, hi=14)
    pk = _host.parse_report(part)
    sr = ShardedRecruiter(0, lib=lib, rank=rank, world=world, rendezvous=rdv)
    sr.engine.set_param("dist_slots", 2048); sr.engine.set_param("dist_block", 128)
    sr.load(pk, 1)
    out = sr.run(edge_cap=400000, **P)
    edges = sr.engine.edges(out["local_edges"])
    chk = cport.edge_checksum(edges)
    total_chk = int(sum(int(x) for x in sr.allreduce([chk & 0xFFFFFFFF, chk >> 32], "sum") * np.array([1, 2 ** 32], dtype=object)) % 2 ** 64)
    n_odd = int(sr.allreduce([int(((edges[:, 1] >= sr.kset.n_acgt) | (edges[:, 2] >= sr.kset.n_acgt)).sum())], "sum")[0])
    if rank == 0:
        pf = _host.parse_report(full)
        with Engine(0, lib) as e:
            e.set_param("dist_slots", 2048); e.set_param("dist_block", 128)
            e.load(pf, 1)
            e.count_kmers(P["k"]); e.select_rare(P["max_nonuniq"], P["lo"], P["hi"])
            kset = km.KmerSet(e.kmers(), P["k"], pf.exotic_rare(P["k"], P["max_nonuniq"], P["lo"], P["hi"]))
            e.set_kmers(kset.codes, P["k"])
            e.build_clouds()
            e.set_clouds(*add_exotic_entries(pf, 1, kset, *e.clouds()))
            e.reset_unique()
            n = e.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], 0, 1, 400000)
            want = dict(n_edges=n, checksum=cport.edge_checksum(e.edges(n)), unique=e.unique_mask(), st=e.stats())
        ok = dict(n_extra=len(kset.extra), same_strings=list(sr.kset.extra) == list(kset.extra), same_codes=bool(np.array_equal(sr.kset.codes, kset.codes)),
                  n_rare=out["n_rare"] == len(kset), n_edges=out["n_edges"] == want["n_edges"], edge_checksum=total_chk == want["checksum"],
                  unique=bool(np.array_equal(sr.unique_mask, want["unique"])), n_emissions=out["n_emissions"] == want["st"]["n_emissions"],
                  n_cloud_entries=out["n_cloud_entries"] == want["st"]["n_cloud_entries"], edges_with_such_a_kmer=n_odd)
        print("EXOTIC_REAL " + json.dumps(ok), flush=True)
    sr.barrier()
    sr.close()


def main():
    rank, world, rdv = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    lib = _lib.load(os.path.join(ROOT, "tests", "emu", "libcfhip_emu.so"))
    pk = _host.synth(n_reads=READS_PER_RANK, cand_offset=rank, cand_stride=world, **SYNTH)
    sr = ShardedRecruiter(0, lib=lib, rank=rank, world=world, rendezvous=rdv)
    sr.engine.set_param("dist_slots", 2048); sr.engine.set_param("dist_block", 128)
    if len(sys.argv) > 4:       # many small rounds per pair of ranks (the loop of cf_comm.h that RCCL runs in 256 MB rounds)
        sr.engine.set_param("comm_round_bytes", int(sys.argv[4]))
    if len(sys.argv) > 5:
        sr.engine.set_param("comm_self_p2p", int(sys.argv[5]))
    sr.load(pk, 1)
    if len(sys.argv) > 6 and sys.argv[6] == "exotic_real":
        return exotic_real(rank, world, rdv, lib)
    if len(sys.argv) > 6:       # exotic windows (symbols other than upper-case ACGT): the decision is made on counts ADDED over the shards
        mode = sys.argv[6]

        class WithExotic:       # the rank's pack + crafted rows of cfh_exotic_list: {h1, h2, pres, multi, no lower-case letter}
            def __init__(self, inner, rows):
                self._inner, self._rows, self.non_acgt = inner, np.array(rows, np.int64).reshape(-1, 5), bool(len(rows))

            def exotic_list(self, k):
                return self._rows

            def __getattr__(self, name):
                return getattr(self._inner, name)
        W, X = [11, 22], [33, 44]
        rows = {"exotic_block": {0: [W + [2, 0, 1], X + [1, 0, 1]], 1: [W + [3, 1, 1]]},        # W: pres 2 + 3 = 5 in [4, 14], multi 1 <= 3, upper case: rare
                "exotic_pass": {0: [W + [1, 0, 1]], 1: [X + [2, 0, 1], W + [9, 4, 1]]},          # W: pres 10 but multi 4 > 3; X: pres 2 < 4: nothing rare
                "exotic_lower": {0: [W + [3, 0, 0]], 1: [W + [3, 0, 0]]}}[mode].get(rank, [])  # rare, but holds a lower-case letter: cannot match an upper-cased unit
        sr.packed = WithExotic(pk, rows)
        try:
            sr.run(edge_cap=200000, **PARAMS)
            refused = False
        except ValueError as e:
            refused = "rare k-mer" in str(e)
        print("EXOTIC_RESULT " + json.dumps(dict(rank=rank, mode=mode, refused=refused)), flush=True)
        sr.close()
        return
    outs = [sr.run(edge_cap=200000, **PARAMS) for _ in range(2)]       # two steps: the local table is rebuilt from the shard
    out = outs[-1]
    edges = sr.engine.edges(out["local_edges"])
    chk = cport.edge_checksum(edges)
    total_chk = int(sum(int(x) for x in sr.allreduce([chk & 0xFFFFFFFF, chk >> 32], "sum") * np.array([1, 2 ** 32], dtype=object)) % 2 ** 64)
    if rank == 0:
        packs = [_host.synth(n_reads=READS_PER_RANK, cand_offset=r, cand_stride=world, **SYNTH) for r in range(world)]
        bases = np.concatenate([p.bases for p in packs])
        read_off = np.concatenate([[0]] + [p.read_off[1:] + off for p, off in zip(packs, np.cumsum([0] + [p.n_bases for p in packs[:-1]]))])
        uss, ues = [], []
        boff = uoff = 0
        up_all = [0]
        for p in packs:
            up, us, ue, _ = p.units(1)
            up_all.extend((up[1:] + uoff).tolist()); uss.append(us + boff); ues.append(ue + boff)
            boff += p.n_bases; uoff += us.size
        c, a = cport.stage2(bases, read_off, np.array(up_all), np.concatenate(uss), np.concatenate(ues), PARAMS["k"], PARAMS["max_nonuniq"],
                            PARAMS["lo"], PARAMS["hi"], 0, 2 ** 62, PARAMS["min_d"], PARAMS["max_d"], PARAMS["min_cov"], PARAMS["rel_threshold"],
                            want_arrays=True)
        keys = ("n_edges", "n_emissions", "n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_cloud_entries")
        ok = dict(
            rare=bool(np.array_equal(sr.rare, a["rare"])),
            unique=bool(np.array_equal(sr.unique_mask, a["unique"])),
            counters=all(out[k] == c[k] for k in keys),
            steps_identical=all(outs[0][k] == out[k] for k in keys + ("n_rare", "n_unique")),
            edge_checksum=total_chk == c["edge_checksum"], n_edges=c["n_edges"], n_rare=c["n_rare"], world=world,
            exchange_bytes=sr.exchange_bytes, got={k: out[k] for k in keys}, want={k: c[k] for k in keys})
        print("SHARDED_RESULT " + json.dumps(ok), flush=True)
    sr.barrier()
    sr.close()


if __name__ == "__main__":
    main()
