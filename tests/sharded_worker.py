"""Worker of tests/test_sharded_gloo.py: one rank of the sharded recruit+distance path on the
host-emulated kernels (CPU tensors, gloo).  Rank 0 checks the result against the C oracle run on
the union of all shards."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

from centroflye_amd import _host, _lib
from centroflye_amd import sharded
from centroflye_amd.sharded import ShardedRecruiter
from oracle import cport

SYNTH = dict(seed=17, unit_len=200, monomer_len=50, n_units=60, flank=60000, mean_len=6500, sigma=0.2, min_len=6000,
             max_len=7000, unit_div=0.03, n_prefix=2, n_suffix=2, prefix_threshold=50000)
PARAMS = dict(k=19, max_nonuniq=3, lo=4, hi=14, min_d=1, max_d=2, min_cov=2, rel_threshold=0.8)
READS_PER_RANK = 12


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.load(os.path.join(ROOT, "tests", "emu", "libcfhip_emu.so"))
    pk = _host.synth(n_reads=READS_PER_RANK, cand_offset=rank, cand_stride=world, **SYNTH)
    sharded.CHUNK_BYTES = 4096      # many rounds per exchange: the multi-round path of the all-to-all / all-gathers
    sr = ShardedRecruiter(0, lib=lib, torch_device="cpu")
    sr.local.set_param("dist_slots", 2048); sr.local.set_param("dist_block", 128)
    sr.glob.set_param("dist_slots", 2048); sr.glob.set_param("dist_block", 128)
    sr.load(pk, 1)
    out = sr.run(edge_cap=200000, **PARAMS)
    edges = sr.dist_engine.edges(out["local_edges"])
    chk = torch.tensor([cport.edge_checksum(edges) % 2 ** 62, (cport.edge_checksum(edges) >> 62)], dtype=torch.int64)
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    if rank == 0:
        total_chk = sum(int(c[0]) + (int(c[1]) << 62) for c in allc) % 2 ** 64
        packs = [_host.synth(n_reads=READS_PER_RANK, cand_offset=r, cand_stride=world, **SYNTH) for r in range(world)]
        bases = np.concatenate([p.bases for p in packs])
        read_off = np.concatenate([[0]] + [p.read_off[1:] + off for p, off in zip(packs, np.cumsum([0] + [p.n_bases for p in packs[:-1]]))])
        ups, uss, ues = [], [], []
        boff = uoff = 0
        up_all = [0]
        for p in packs:
            up, us, ue, _ = p.units(1)
            up_all.extend((up[1:] + uoff).tolist()); uss.append(us + boff); ues.append(ue + boff)
            boff += p.n_bases; uoff += us.size
        c, a = cport.stage2(bases, read_off, np.array(up_all), np.concatenate(uss), np.concatenate(ues), PARAMS["k"], PARAMS["max_nonuniq"],
                            PARAMS["lo"], PARAMS["hi"], 0, 2 ** 62, PARAMS["min_d"], PARAMS["max_d"], PARAMS["min_cov"], PARAMS["rel_threshold"],
                            want_arrays=True)
        ok = dict(
            rare=bool(np.array_equal(sr.rare, a["rare"])),
            unique=bool(np.array_equal(sr.unique_mask, a["unique"])),
            counters=all(out[k] == c[k] for k in ("n_edges", "n_emissions", "n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_cloud_entries")),
            edge_checksum=total_chk == c["edge_checksum"], n_edges=c["n_edges"], n_rare=c["n_rare"], world=world)
        print("SHARDED_RESULT " + json.dumps(ok), flush=True)
    dist.barrier()
    sr.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
