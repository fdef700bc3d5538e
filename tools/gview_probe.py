#!/usr/bin/env python3
"""Developer tool (GPU box): one rank-of-N partition of the distance stage at BASELINE configs[3]'s size, four ways — over the local
clouds or over the gathered view (cf_allgather_clouds through a one-rank communicator), with a small edge cap or with every selected
edge stored — to tell apart what made the partition of profiles/r05_parity_500k_rank3.json take 1 368 ms where the emulated rank of
round 4 (local clouds, 2^20 edges stored) took 186 ms.
With --shapes: the same partition (local clouds, 2^20 edges stored) in other launch shapes of cf_dist_kernel (workgroups per CU x threads):
a rank of 8 at this size has 11 000 pair emissions per first k-mer, half of the single-GPU bench's.
usage: tools/gview_probe.py [reads=500000] [part=3] [n_parts=8] [--shapes] [--lib centroflye_amd/build_variants/stamps.so]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host
from centroflye_amd.engine import Engine
shapes = "--shapes" in sys.argv
sys.argv = [a for a in sys.argv if a != "--shapes"]
lib_path = None      # --lib <build>: another build of the library (a -DCF_DIST_STAMPS one prints its phase shares); only the first case is run then
if "--lib" in sys.argv:
    i = sys.argv.index("--lib"); lib_path = sys.argv[i + 1]; del sys.argv[i:i + 2]
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
part = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n_parts = int(sys.argv[3]) if len(sys.argv) > 3 else 8
P = dict(k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)
pk = _host.synth(n_reads=reads, seed=4, n_units=max(24, int(round(0.3 * reads))), var_len=8)
from centroflye_amd import _lib
e = Engine(0, _lib.load(os.path.join(ROOT, lib_path)) if lib_path else None)
e.load(pk, 1); e.count_kmers(P["k"]); e.select_rare(P["max_nonuniq"], P["lo"], P["hi"]); e.build_clouds()
out = []


def run(tag, cap):
    for rep in range(2):
        e.reset_unique()
        t0 = time.perf_counter()
        ne = e.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], part, n_parts, cap)
        dt = time.perf_counter() - t0
        st, tm = e.stats(), e.times()
        rec = dict(case=tag, rep=rep, edge_cap=cap, n_edges=ne, n_emissions=st["n_emissions"], kernel_ms=round(tm["dist_kernel_ms"], 1), setup_ms=round(tm["postings_ms"], 1),
                   call_ms=round(dt * 1e3, 1), hbm_live_gb=round(st["hbm_bytes_live"] / 1e9, 1))
        out.append(rec); print(json.dumps(rec), flush=True)
    return ne


n = run("local clouds, 2^20 edges stored", 1 << 20)
if lib_path:
    e.close(); sys.exit(0)
if shapes:
    for wgs, block in ((2, 512), (3, 320), (4, 256), (2, 384), (1, 1024)):
        e.set_param("dist_wgs", wgs); e.set_param("dist_block", block)
        try:
            run(f"local clouds, 2^20 edges stored, {wgs} workgroups of {block} threads per CU", 1 << 20)
        except Exception as ex:      # a shape the LDS carve-up does not allow
            print(json.dumps(dict(case=f"{wgs} x {block}", error=str(ex)[:200])), flush=True)
    e.set_param("dist_wgs", 0); e.set_param("dist_block", 0)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gview_probe_shapes.json"), "w"), indent=1)
    e.close(); sys.exit(0)
run("local clouds, every edge stored", n + 16)
run("local clouds, 2^20 edges stored (again, after the large edge buffer)", 1 << 20)
e.set_param("comm_self_p2p", 1)
e.comm_init(0, 1, os.path.join(os.environ.get("TMPDIR", "/tmp"), f"cf_gview_probe_{os.getpid()}.id"))
e.allgather_clouds()
run("gathered view, 2^20 edges stored", 1 << 20)
run("gathered view, every edge stored", n + 16)
e.comm_free(); e.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gview_probe.json"), "w"), indent=1)
