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


class ShardedRecruiter:
    def __init__(self, device_index=0, lib=None, rank=0, world=1, rendezvous=None, force_exchange=False):
        self.rank, self.world = int(rank), int(world)
        # force_exchange: run the exchange path (bucketing, all-to-all, all-gathers, gathered view) even with one rank —
        # used by the tests to exercise the device-side exchange kernels and the RCCL plumbing on a single GPU
        self.exchange = self.world > 1 or bool(force_exchange)
        self.engine = Engine(device_index, lib)
        self.local = self.glob = self.dist_engine = self.engine     # one context per GPU
        if self.exchange:
            # every rank of this launch sees the same nonce (they share the launcher): a rendezvous file left by another launch under a
            # reused name is not taken for this one's (cf_comm_rccl.hip)
            os.environ.setdefault("CF_COMM_NONCE", os.path.basename(default_rendezvous()))
            self.engine.comm_init(self.rank, self.world, rendezvous if rendezvous is not None else default_rendezvous())
        self.sections = {}
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
        E = self.engine
        sec, t_last = {}, [time.perf_counter()]

        def lap(name):      # every library call returns with its work done
            now = time.perf_counter()
            sec[name] = round(sec.get(name, 0.0) + now - t_last[0], 4)
            t_last[0] = now
        self.sections = sec
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
            cloud_ptr, entries = add_exotic_entries(packed, 1, self.kset, *E.clouds())
            E.set_clouds(cloud_ptr, entries)
            n_ce_local = int(entries.size)
        lap("clouds")
        dist_ce = n_ce_local
        if self.exchange:
            dist_ce = E.allgather_clouds()
            lap("cloud_gather")
        E.reset_unique()
        n_edges = E.dist_edges(0, 2 ** 62, min_d, max_d, min_cov, rel_threshold, self.rank, self.world, edge_cap)
        st_d = E.stats()
        lap("dist")
        c = [n_edges, st_d["n_emissions"], st_local["n_bases"], st_local["n_windows"], st_local["n_read_kmers"],
             st_owner["n_distinct"], st_owner["n_kept"], n_ce_local]
        if self.exchange:
            n_unique = E.allreduce_unique()
            c = E.allreduce(c, "sum").tolist()
        else:
            n_unique = st_d["n_unique"]
        lap("combine")
        self.local_edges = n_edges
        return dict(n_edges=c[0], n_emissions=c[1], n_bases=c[2], n_windows=c[3], n_read_kmers=c[4], n_distinct=c[5],
                    n_kept=c[6], n_cloud_entries=c[7], n_rare=n_rare, n_unique=n_unique,
                    local_emissions=st_d["n_emissions"], local_edges=n_edges, local_bases=st_local["n_bases"],
                    local_cloud_entries=n_ce_local, dist_cloud_entries=dist_ce,
                    dist_kernel_ms=E.times()["dist_kernel_ms"], n_spilled=st_d["n_spilled"], n_dist_passes=st_d["n_dist_passes"])
