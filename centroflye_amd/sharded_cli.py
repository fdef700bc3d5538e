"""Stage 2 on N GPUs behind the reference's own command line.

``scripts/distance_based_kmer_recruitment.py`` (reference :174-208; spawned by ``centroFlye.py:172-192``) is a single process.
Here the same command line, with ``CF_GPUS=N`` in the environment (or under a launcher that exports ``RANK`` / ``WORLD_SIZE``:
``python -m torch.distributed.run --nproc-per-node N scripts/distance_based_kmer_recruitment.py ...``), runs the sharded path of
``centroflye_amd/sharded.py`` and writes the reference's two files:

  parent (``launch``)   never touches a GPU: parses the report ONCE on the host into the binary pack cache (``CF_PACK_CACHE``; a
                        private directory under the output directory when unset), starts the N ranks as child processes — before
                        any GPU call, never re-executing a process that has initialised HIP — and fails fast when one dies;
  rank (``rank_main``)  maps the pack, takes a contiguous run of reads holding 1 / N of the bases (so the gathered clouds keep the
                        record order and --min-nreads / --max-nreads mean what they mean on one GPU), runs
                        ``ShardedRecruiter.prepare`` (A1, table exchange, A2, gathers, A3) and then A5 + A6 on its first k-mers
                        a % N == rank in sub-partitions small enough to fetch, writing its sorted edge rows to a part file;
  rank 0                writes ``unique_kmers_min_edge_cov_{N}.txt`` from the OR of all ranks' masks, appends the part files in rank
                        order to ``unique_edges_min_edge_cov_{N}.txt`` (``--no-edges`` skips them), and the metrics file.

The k-mer file is byte-identical to the single-GPU one (and to the reference's); the edge file holds the same lines in another
order (the reference's order is its dicts' insertion order).
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np


def launch(argv, n_gpus, rank_cmd=None):
    """Parent of a CF_GPUS=N run.  rank_cmd: the command that runs ONE rank (tests start their own worker, which binds the
    host-emulated kernels; the default is this package's stage script)."""
    from . import _host
    from . import distance_based_kmer_recruitment as D
    from .sharded import new_launch_token
    params = D.parse_args(argv)
    D.smart_makedirs(params.outdir)
    env = dict(os.environ)
    private_cache = idf = None
    procs = []
    rc = 1
    try:      #  whatever happens below — an unreadable report, an interrupt — no rank and no private file is left behind
        if not env.get("CF_PACK_CACHE"):
            private_cache = tempfile.mkdtemp(prefix=".cfpack_", dir=params.outdir)
            env["CF_PACK_CACHE"] = private_cache
        t0 = time.time()
        pk = _host.parse_report(params.ncrf, keep_rows=False, cache_dir=env["CF_PACK_CACHE"])      # host only; writes the pack the ranks map
        n_reads = pk.n_reads
        del pk
        if params.verbose:
            print(f"# {n_reads} reads parsed in {time.time() - t0:.2f} s; starting {n_gpus} ranks")
        idf = os.path.join(tempfile.gettempdir(), f"cfcomm_{os.getuid()}_{os.getpid()}_{int(time.time() * 1e3)}.id")
        env.update(WORLD_SIZE=str(n_gpus), CF_COMM_ID_FILE=idf, CF_COMM_NONCE=new_launch_token(), CF_GPUS="")
        # The ranks run the stage script ITSELF, by path (`python -m centroflye_amd...` only resolves when the working
        # directory is the repository — centroFlye.py:172-188 calls the script by absolute path from the user's directory); the package's
        # parent directory goes in front of the children's PYTHONPATH as well, for a caller that imported the module some other way.
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        script = os.path.join(pkg_parent, "scripts", "distance_based_kmer_recruitment.py")
        default_cmd = [sys.executable, script] if os.path.exists(script) else [sys.executable, "-m", "centroflye_amd.distance_based_kmer_recruitment"]
        cmd = list(rank_cmd) if rank_cmd else default_cmd
        args = list(sys.argv[1:] if argv is None else argv)
        procs = [subprocess.Popen(cmd + args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n_gpus)]
        rc, live = 0, list(procs)
        while live and rc == 0:      # a rank that dies leaves the others waiting in a collective: end them, exit non-zero
            time.sleep(0.1)
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = max(rc, abs(code))
        if rc:
            print(f"distance_based_kmer_recruitment: a rank exited with code {rc}; the other ranks are stopped", file=sys.stderr)
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t_end = time.time() + 10
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        for leftover in (idf, private_cache):
            if leftover and os.path.isdir(leftover):
                shutil.rmtree(leftover, ignore_errors=True)
            elif leftover and os.path.exists(leftover):
                os.remove(leftover)
    return rc


class ReadShard:
    """Reads [lo, hi) of a pack with the attributes the device hand-over and the exotic-window side path read
    (Engine.load, ShardedRecruiter.prepare, read_kmer_cloud.exotic_hits) — views, no copies."""

    def __init__(self, packed, lo, hi):
        self._pk, self.lo, self.hi = packed, int(lo), int(hi)
        ro = packed.read_off
        self._b0 = int(ro[lo])
        self.n_reads = self.hi - self.lo
        self.read_off = ro[lo:hi + 1] - self._b0
        self.bases = packed.bases[self._b0:int(ro[hi])]
        self.n_bases = int(self.bases.size)
        self.ids = packed.ids[lo:hi]
        self.motifs = packed.motifs
        # (conservative: some shard of a pack with such symbols holds them; a shard without any lists no window)
        self.non_acgt = bool(packed.non_acgt)

    def units(self, n=1):
        up, us, ue, uc = self._pk.units(n)
        u0, u1 = int(up[self.lo]), int(up[self.hi])
        return up[self.lo:self.hi + 1] - u0, us[u0:u1] - self._b0, ue[u0:u1] - self._b0, uc[u0:u1]

    def exotic_list(self, k):
        return self._pk.exotic_list(k, self.lo, self.hi)

    def exotic_rare(self, k, max_nonuniq, lo, hi):
        # only ever asked for "every window without a lower-case letter, whatever its counts" (the text behind a pair of hashes):
        # the whole pack's windows are a superset of the shard's
        return self._pk.exotic_rare(k, max_nonuniq, lo, hi)


def shard_bounds(read_off, world):
    """Contiguous runs of reads holding about 1 / world of the bases each: bounds[r] .. bounds[r + 1] is rank r's."""
    read_off = np.asarray(read_off, np.int64)
    n_b = int(read_off[-1])
    cuts = [int(np.searchsorted(read_off, (n_b * r) // world, side="left")) for r in range(world)] + [read_off.size - 1]
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts


def rank_main(argv=None, lib=None, device=None, sub_edges=None, knobs=None):
    """One rank.  lib / device / sub_edges / knobs: test hooks (another build of the device library, a device index other than
    LOCAL_RANK, the edge rows fetched per sub-partition, cf_set_param settings)."""
    from . import _host
    from . import distance_based_kmer_recruitment as D
    from . import kmers as km
    from .sharded import ShardedRecruiter
    params = D.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if "LOCAL_RANK" not in os.environ and device is None and os.environ.get("CF_SINGLE_NODE", "1") == "0":
        #  RANK is the device index only on one node; a multi-node launcher that does not export LOCAL_RANK must say so
        raise SystemExit("distance_based_kmer_recruitment: WORLD_SIZE spans several nodes (CF_SINGLE_NODE=0) but LOCAL_RANK is not set")
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    D.smart_makedirs(params.outdir)
    t0 = time.time()
    pk = _host.parse_report(params.ncrf, keep_rows=False)
    t_parse = time.time() - t0
    cuts = shard_bounds(pk.read_off, world)
    shard = ReadShard(pk, cuts[rank], cuts[rank + 1])
    sr = ShardedRecruiter(local_rank if device is None else device, lib=lib, rank=rank, world=world, force_exchange=True)
    try:
        for name, value in (knobs or {}).items():
            sr.engine.set_param(name, value)
        sr.load(shard, 1)
        lo, hi = D.rare_window(params.bottom, params.top, params.coverage, params.kmer_survival_rate)
        t1 = time.time()
        n_rare = sr.prepare(params.k, params.max_nonuniq, lo, hi)
        kset = sr.kset if sr.kset is not None else km.KmerSet(sr.rare, params.k, [])
        if params.verbose and rank == 0:
            print(f"# rare kmers: {n_rare}")
        want_edges = not params.no_edges
        efile = os.path.join(params.outdir, f"unique_edges_min_edge_cov_{params.min_coverage}.txt")
        part_fn = f"{efile}.rank{rank}.tmp"
        chunk = int(sub_edges or D.EDGE_CHUNK)
        n_sub = 1
        while True:      # this rank's first k-mers in sub-partitions small enough to fetch (as the single-GPU script does)
            sr.reset_dist()
            if want_edges:
                open(part_fn, "w").close()
            again = False
            for sub in range(n_sub):
                n = sr.dist(params.min_distance, params.max_distance, params.min_coverage, 0.8, chunk if want_edges else 0, sub, n_sub,
                            params.min_nreads, params.max_nreads)
                if not want_edges:
                    continue
                if n > chunk:
                    # (more sub-partitions — but never more than there are first k-mers: a single first k-mer with more edges than the chunk
                    # holds, which only a test's tiny chunk makes possible, gets a larger chunk instead; round 5, tools/fuzz_cli_vs_reference.py:
                    # the count grew until world * n_sub left the 32-bit range)
                    if n_sub >= max(1, n_rare):
                        chunk = int(n)
                    else:
                        n_sub = min(max(2 * n_sub, int(n_sub * (n / chunk) * 1.5) + 1), max(1, n_rare))
                    again = True
                    break
                sr.engine.sort_edges()
                D.write_edge_file(part_fn, kset, sr.engine.edges(n), append=True)
            if not again:
                break
        res = sr.combine()
        t_dist = time.time() - t1
        mask = sr.unique_mask
        sr.barrier()      # every part file is complete
        if rank == 0:
            kfile = os.path.join(params.outdir, f"unique_kmers_min_edge_cov_{params.min_coverage}.txt")
            D.write_kmer_file(kfile + ".tmp", kset, np.flatnonzero(mask))
            os.replace(kfile + ".tmp", kfile)
            if want_edges:
                with open(efile + ".tmp", "wb") as out:
                    missing = [r for r in range(world) if not os.path.exists(f"{efile}.rank{r}.tmp")]
                    if missing:      #  every rank writes its part into --outdir: ranks on other nodes need it on a shared file system
                        raise SystemExit(f"distance_based_kmer_recruitment: the edge parts of ranks {missing} are not in {params.outdir} — "
                                         "with ranks on several nodes --outdir must be on a file system all of them share")
                    for r in range(world):
                        fn = f"{efile}.rank{r}.tmp"
                        with open(fn, "rb") as f:
                            shutil.copyfileobj(f, out, 1 << 24)
                        os.remove(fn)
                os.replace(efile + ".tmp", efile)
            if params.verbose:
                print(f"# unique kmers: {int(mask.sum())}; edges: {res['n_edges']}")
            if params.metrics:
                with open(os.path.join(params.outdir, "stage2_metrics.json"), "w") as f:
                    json.dump(dict(stats=res, world=world, shard_reads=[cuts[r + 1] - cuts[r] for r in range(world)], sections_s=sr.sections,
                                   exchange_bytes=int(sr.exchange_bytes), parse_s=t_parse, dist_wall_s=t_dist, total_s=time.time() - t0,
                                   pack_from_cache=bool(getattr(pk, "from_cache", False)), device=sr.engine.device_info()), f, indent=1)
        sr.barrier()
    finally:
        sr.close()
    return 0
