"""Multi-GPU recruit + distance (A1-A6): one process per GPU, reads sharded across ranks.

The reference has no distributed code at all; this is new design following
SURVEY.md §8(e).  Every step below is ONE call into libcfhip.so (include/cfhip.h); the exchange
kernels and the RCCL calls live inside the library (csrc/hip/cf_exchange.hip, cf_comm_rccl.hip),
this module only sequences them — no torch, no tensors:

  1. cf_count_kmers        A1 on the local read shard                          (no traffic)
  2. cf_exchange_table     records bucketed by hash(key) % n on the device, one all-to-all,
                           the owner adds what it receives        -> exact global counts
  3. cf_select_rare        A2 on the owned keys
     cf_allgather_kmers    all-gather of the rare lists, sorted union installed everywhere
  4. cf_build_clouds       A3 on the local shard
     cf_allgather_clouds   all-gather of the per-unit clouds (CSR)
  5. cf_dist_edges         A5+A6 on first k-mers a with a % n == rank over ALL clouds — no reduction
  6. cf_allreduce_unique   OR of the unique-k-mer masks; counters summed with cf_comm_allreduce_i64
"""
import os
import time

import numpy as np

from . import kmers as km
from .engine import Engine
from .read_kmer_cloud import add_exotic_entries


def default_rendezvous():
    """A path all ranks of one launch share and no other launch uses: the launcher's (= the common parent's) pid and
    the master port tell launches apart.  CF_COMM_ID_FILE overrides it."""
    p = os.environ.get("CF_COMM_ID_FILE")
    if p:
        return p
    ppid, born = os.getppid(), "0"
    try:        # the launcher's start time (clock ticks since boot): a recycled pid cannot collide with a stale file
        with open(f"/proc/{ppid}/stat") as f:
            born = f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    return os.path.join(os.environ.get("TMPDIR", "/tmp"),
                        f"cfcomm_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}_{ppid}_{born}.id")


def new_launch_token():
    """A fresh token for CF_COMM_NONCE: a launcher (bench.py --gpus N, the CF_GPUS entry of the stage script) exports it to every rank
    it starts, so that a rendezvous file left by another launch under a reused name is never taken for this one's."""
    return os.urandom(8).hex()


def set_launch_nonce(explicit_rendezvous):
    """CF_COMM_NONCE for cf_comm_init (cf_comm_rccl.hip), from values that are EQUAL on all ranks of this launch and FRESH per launch
     — or left unset (no nonce check; the record's age rule still applies) when no such value is known:
      * a launcher's token (already in the environment): kept;
      * the ranks share a parent and the rendezvous name is derived from it (no CF_COMM_ID_FILE, no explicit path): that name, which
        holds the parent's pid and start time;
      * an explicit rendezvous / CF_COMM_ID_FILE (mpirun, srun, one launcher per node: parents differ, the name may be reused):
        the job's id as the launcher publishes it (TORCHELASTIC_RUN_ID other than torchrun's default "none", SLURM_JOB_ID + SLURM_STEP_ID,
        OMPI / PMIx job ids), else nothing."""
    if os.environ.get("CF_COMM_NONCE"):
        return
    if explicit_rendezvous is None and not os.environ.get("CF_COMM_ID_FILE"):
        os.environ["CF_COMM_NONCE"] = os.path.basename(default_rendezvous())
        return
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
    if run_id and run_id != "none":
        os.environ["CF_COMM_NONCE"] = f"torchelastic:{run_id}:{os.environ.get('MASTER_ADDR', '')}:{os.environ.get('MASTER_PORT', '')}"
    elif os.environ.get("SLURM_JOB_ID"):
        os.environ["CF_COMM_NONCE"] = f"slurm:{os.environ['SLURM_JOB_ID']}:{os.environ.get('SLURM_STEP_ID', '')}"
    elif os.environ.get("PMIX_NAMESPACE") or os.environ.get("OMPI_MCA_ess_base_jobid"):
        os.environ["CF_COMM_NONCE"] = "mpi:" + (os.environ.get("PMIX_NAMESPACE") or os.environ["OMPI_MCA_ess_base_jobid"])


class ShardedRecruiter:
    def __init__(self, device_index=0, lib=None, rank=0, world=1, rendezvous=None, force_exchange=False):
        self.rank, self.world = int(rank), int(world)
        # force_exchange: run the exchange path (bucketing, all-to-all, all-gathers, gathered view) even with one rank —
        # used by the tests to exercise the device-side exchange kernels and the RCCL plumbing on a single GPU
        self.exchange = self.world > 1 or bool(force_exchange)
        self.engine = Engine(device_index, lib)
        self.local = self.glob = self.dist_engine = self.engine     # one context per GPU
        if self.exchange:
            set_launch_nonce(rendezvous)
            self.engine.comm_init(self.rank, self.world, rendezvous if rendezvous is not None else default_rendezvous())
        self.sections, self._t_last = {}, time.perf_counter()
        self.exchange_bytes = 0
        self._cache = {}

    # results of the last run(): device resident (HBM) until read, like the inputs
    @property
    def unique_mask(self):
        """bool[n_rare]: rare k-mers selected by some edge (all ranks' edges when sharded)."""
        if "mask" not in self._cache:
            self._cache["mask"] = self.engine.unique_mask()
        return self._cache["mask"]

    @property
    def rare(self):
        """uint64[n_rare]: the rare k-mer set (2-bit codes, ascending)."""
        if "rare" not in self._cache:
            self._cache["rare"] = self.engine.kmers()
        return self._cache["rare"]

    def close(self):
        self.engine.close()

    def load(self, packed, n_motif=1):
        self.packed = packed
        self.n_motif = int(n_motif)
        self.engine.load(packed, n_motif)

    def barrier(self):
        if self.exchange:
            self.engine.allreduce([0], "sum")

    def allreduce(self, values, op="sum"):
        return self.engine.allreduce(values, op) if self.exchange else np.asarray(values, np.int64)

    def _exotic_strings(self, packed, k, rare_keys):
        """The text of the windows whose (hash, hash) rows turned out rare over ALL shards: every rank looks its own windows up and the
        ranks exchange the bytes (7 per int64, so that the sum of the one rank that fills a word stays its value) — each window is held
        by at least one shard."""
        want = {(int(a), int(b)): i for i, (a, b) in enumerate(rare_keys.tolist())}
        words = (k + 6) // 7
        slab = np.zeros((len(want), 1 + words), np.int64)
        for s_ in packed.exotic_rare(k, 2 ** 31 - 1, 0, 2 ** 32 - 1):      # this shard's windows without a lower-case letter, whatever their counts
            raw = s_.encode("latin-1")
            h1, h2 = 0xcbf29ce484222325, 0x9E3779B97F4A7C15      # the two hashes of cfh_exotic_list (cfhost.cpp)
            for c in raw:
                h1 = ((h1 ^ c) * 0x100000001b3) & (2 ** 64 - 1)
                h2 = ((h2 + c) * 0xff51afd7ed558ccd) & (2 ** 64 - 1)
                h2 ^= h2 >> 29
            i = want.get((h1 >> 1, h2 >> 1))
            if i is not None:
                slab[i, 0] = 1
                for w in range(words):
                    slab[i, 1 + w] = int.from_bytes(raw[7 * w:7 * w + 7].ljust(7, b"\0"), "big")
        if self.exchange and self.world > 1:
            slab = np.asarray(self.allreduce(slab.reshape(-1), "sum"), np.int64).reshape(len(want), 1 + words)
        if (slab[:, 0] < 1).any():
            raise ValueError(f"{int((slab[:, 0] < 1).sum())} rare k-mer window(s) with a symbol other than A, C, G, T: their text was found on no shard")
        out = []
        for row in slab.tolist():
            n = row[0]
            out.append(b"".join(int(v // n).to_bytes(7, "big") for v in row[1:])[:k].decode("latin-1"))      # (held by n shards: n equal words added)
        return sorted(out)

    # ------------------------------------------------------------------ one step
    def run(self, k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8, edge_cap=0):
        """One step: prepare() (A1-A3 with their exchanges), one distance launch on this rank's first k-mers, combine()."""
        self.prepare(k, max_nonuniq, lo, hi)
        self.dist(min_d, max_d, min_cov, rel_threshold, edge_cap)
        return self.combine()

    def _lap(self, name):      # every library call returns with its work done
        now = time.perf_counter()
        self.sections[name] = round(self.sections.get(name, 0.0) + now - self._t_last, 4)
        self._t_last = now

    def prepare(self, k=19, max_nonuniq=3, lo=10, hi=32):
        """A1 on the shard, table exchange, A2 on the owned keys, rare-list gather, A3 on the shard, cloud gather: afterwards every rank
        holds the rare set and the clouds of ALL reads (units in rank order) and dist() may run any first-k-mer partition."""
        E = self.engine
        self.sections, self._t_last = {}, time.perf_counter()
        lap = self._lap
        self._cache = {}
        E.count_kmers(k)
        st_local = E.stats()
        lap("count")
        # Windows with symbols other than upper-case A, C, G, T are skipped on the device; the reference counts them as k-mers of
        # their own (distance_based_kmer_recruitment.py:47-53).  One of them changes an output only if it is rare over ALL
        # shards and could match a window of an upper-cased unit: decided on GLOBAL counts, exactly as the single-GPU entry
        # point decides (check_exotic_windows) — presence and multiplicity of a window add over disjoint read shards.  Those that
        # are rare travel as strings from here on (round 4; round 3 stopped the run).
        packed = getattr(self, "packed", None)
        self._exotic = []
        if packed is not None:
            rows = packed.exotic_list(k) if packed.non_acgt else np.zeros((0, 5), np.int64)
            # a window's text decides its flag (column 4: no lower-case letter), the same on every shard; windows with a lower-case letter
            # can never be selected (they match no upper-cased unit) — on soft-masked read sets they are nearly all rows, so they are
            # dropped before the exchange
            rows = rows[rows[:, 4] != 0]
            if self.exchange and self.world > 1:
                n_max = int(self.allreduce([rows.shape[0]], "max")[0])
                if n_max:
                    slab = np.zeros((self.world, n_max, 5), np.int64)
                    slab[self.rank, :rows.shape[0]] = rows
                    slab[self.rank, :rows.shape[0], 2] += 1      # (pres + 1 marks a used row: a window's pres is never 0)
                    slab = np.asarray(self.allreduce(slab.reshape(-1), "sum"), np.int64).reshape(self.world * n_max, 5)
                    rows = slab[slab[:, 2] > 0]
                    rows[:, 2] -= 1
                else:
                    rows = np.zeros((0, 5), np.int64)
            extra = []
            if rows.shape[0]:
                keys, inv = np.unique(np.stack([rows[:, 0], rows[:, 1]], 1), axis=0, return_inverse=True)
                inv = np.asarray(inv).reshape(-1)
                pres = np.bincount(inv, weights=rows[:, 2]).astype(np.int64)
                multi = np.bincount(inv, weights=rows[:, 3]).astype(np.int64)
                upper = np.bincount(inv, weights=rows[:, 4]) > 0
                rare_keys = keys[(multi <= max_nonuniq) & (pres >= lo) & (pres <= hi) & upper]
                if rare_keys.shape[0]:
                    extra = self._exotic_strings(packed, k, rare_keys)
            self._exotic = extra
        lap("exotic_windows")
        if self.exchange:
            self.exchange_bytes = E.exchange_table()
            lap("table_exchange")
        E.select_rare(max_nonuniq, lo, hi)
        st_owner = E.stats()
        n_rare = E.allgather_kmers() if self.exchange else st_owner["n_kmers"]
        self.kset = None
        if getattr(self, "_exotic", None):
            # rare k-mers with an N (...): carried as strings beside the codes (kmers.KmerSet.extra, as the single-GPU entry point does);
            # every rank installs the same pseudo-codes behind the gathered list, so ranks agree over all ranks
            self.kset = km.KmerSet(E.kmers(), k, self._exotic)
            E.set_kmers(self.kset.codes, k)
            n_rare = len(self.kset)
        lap("select")
        n_ce_local = E.build_clouds()
        if self.kset is not None:      # their cloud entries (windows over a non-ACGT symbol of this rank's units) come from the host
            cloud_ptr, entries = add_exotic_entries(packed, getattr(self, "n_motif", 1), self.kset, *E.clouds())
            E.set_clouds(cloud_ptr, entries)
            n_ce_local = int(entries.size)
        lap("clouds")
        dist_ce = n_ce_local
        if self.exchange:
            dist_ce = E.allgather_clouds()
            lap("cloud_gather")
        self._prep = dict(st_local=st_local, st_owner=st_owner, n_rare=n_rare, n_ce_local=n_ce_local, dist_ce=dist_ce)
        self.reset_dist()
        return n_rare

    def reset_dist(self):
        """Forget the distance launches since prepare(): empty unique bitmap, counters at zero."""
        self.engine.reset_unique()
        self._dist = dict(n_edges=0, n_emissions=0, kernel_ms=0.0, n_spilled=0, n_dist_passes=0)

    def dist(self, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8, edge_cap=0, sub=0, n_sub=1, min_n=0, max_n=2 ** 62):
        """A5 + A6 for this rank's first k-mers, a % world == rank — or for the sub-partition `sub` of `n_sub` of them
        (a % (world * n_sub) == rank + sub * world: the stage script streams a rank's edges in pieces it can hold).  The unique bitmap
        and the counters accumulate over calls until the next prepare().  Returns the edges selected by this call (up to edge_cap of
        them are stored: Engine.edges / sort_edges)."""
        E = self.engine
        n_edges = E.dist_edges(min_n, max_n, min_d, max_d, min_cov, rel_threshold, self.rank + sub * self.world, self.world * n_sub, edge_cap)
        st_d = E.stats()
        d = self._dist
        d["n_edges"] += n_edges; d["n_emissions"] += st_d["n_emissions"]; d["kernel_ms"] += E.times()["dist_kernel_ms"]
        d["n_spilled"] += st_d["n_spilled"]; d["n_dist_passes"] += st_d["n_dist_passes"]
        d["n_unique_local"] = st_d["n_unique"]
        self._lap("dist")
        return n_edges

    def combine(self):
        """OR of the unique masks and sums of the counters over ranks: the figures of the whole read set."""
        E, p, d = self.engine, self._prep, self._dist
        st_local, st_owner = p["st_local"], p["st_owner"]
        c = [d["n_edges"], d["n_emissions"], st_local["n_bases"], st_local["n_windows"], st_local["n_read_kmers"],
             st_owner["n_distinct"], st_owner["n_kept"], p["n_ce_local"]]
        if self.exchange:
            n_unique = E.allreduce_unique()
            c = E.allreduce(c, "sum").tolist()
        else:
            n_unique = d.get("n_unique_local", 0)
        self._lap("combine")
        self._cache.pop("mask", None)
        self.local_edges = d["n_edges"]
        return dict(n_edges=c[0], n_emissions=c[1], n_bases=c[2], n_windows=c[3], n_read_kmers=c[4], n_distinct=c[5],
                    n_kept=c[6], n_cloud_entries=c[7], n_rare=p["n_rare"], n_unique=n_unique,
                    local_emissions=d["n_emissions"], local_edges=d["n_edges"], local_bases=st_local["n_bases"],
                    local_cloud_entries=p["n_ce_local"], dist_cloud_entries=p["dist_ce"],
                    dist_kernel_ms=d["kernel_ms"], n_spilled=d["n_spilled"], n_dist_passes=d["n_dist_passes"])
