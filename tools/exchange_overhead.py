#!/usr/bin/env python3
"""Developer tool: cost of the multi-GPU exchange path measured on ONE GPU (single-rank RCCL group,
force_exchange): the same code every rank runs at N > 1 minus the wire time.  Prints per-section seconds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from centroflye_amd import _host
from centroflye_amd.sharded import ShardedRecruiter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
P = dict(k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)
for force in (False, True):
    sr = ShardedRecruiter(0, force_exchange=force); sr.load(pk, 1)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = sr.run(edge_cap=1 << 20, **P)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("exchange" if force else "plain", round(dt, 3), getattr(sr, "sections", None), {k: out[k] for k in ("n_edges", "n_emissions", "n_rare", "n_unique", "n_cloud_entries", "n_distinct", "n_kept", "n_dist_passes", "n_spilled")}, flush=True)
    sr.close()
dist.destroy_process_group()
