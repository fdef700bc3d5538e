"""Multi-GPU recruit + distance (A1-A6): one process per GPU, reads sharded across ranks.

The reference has no distributed code at all; this is new design following
SURVEY.md §8(e).  Per rank and step:

  1. A1 on the local read shard                                    (device, no traffic)
  2. the local table is dumped as (key, pres|multi) pairs, bucketed by hash(key) % n and
     exchanged with ONE all-to-all (RCCL send/recv pairs use all xGMI links at once); the
     owner merges what it receives with 64-bit atomic adds         -> exact global counts
  3. A2 on the owned key range, all-gather of the (small) rare lists, every rank sorts the union
  4. A3 on the local shard, all-gather of the per-unit clouds (CSR)
  5. A5+A6 on first k-mers a with a % n == rank over ALL clouds — no reduction needed
  6. all-reduce(max) of the unique-k-mer byte mask, all-reduce(sum) of the counters

torch is used for what it is here for: device buffers for the exchange, ``torch.distributed``
(backend "nccl" = RCCL on ROCm; "gloo" on CPU for the tests) and a few index ops on the exchange
buffers.  All pipeline compute stays in libcfhip behind the C ABI; the buffers are handed over as
raw pointers.  torch must be imported before libcfhip is loaded so that both share one HIP runtime.
"""
import numpy as np
import torch
import torch.distributed as dist

from .engine import Engine

_MUL = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64


def _owner(keys_i64, world):
    h = (keys_i64 ^ (keys_i64 >> 29)) * _MUL
    return ((h >> 33) & 0x7FFFFFFF) % world


# RCCL (2.26, ROCm 7) silently drops part of a large all_to_all_single message (measured on MI355X, tools/
# exchange_overhead.py: 480 MB arrives intact, 1.26 GB loses half its rows), so the table exchange is cut into rounds
# of at most CHUNK_BYTES per call and rank; the all-gathers are cut the same way as a precaution.  Every rank runs the
# same number of rounds (a MAX all-reduce of the counts).
CHUNK_BYTES = 256 << 20


def all_to_all_rows(payload, send_counts, recv_counts, group=None, chunk_rows=None):
    """payload: [sum(send_counts), C] rows grouped by destination rank; returns [sum(recv_counts), C] grouped by source."""
    W = len(send_counts)
    dev = payload.device
    row_bytes = payload.element_size() * (payload.shape[1] if payload.dim() > 1 else 1)
    if chunk_rows is None:
        chunk_rows = max(1, CHUNK_BYTES // (row_bytes * W))
    recv = torch.empty((sum(recv_counts),) + tuple(payload.shape[1:]), dtype=payload.dtype, device=dev)
    most = torch.tensor([max(list(send_counts) + list(recv_counts) + [0])], dtype=torch.int64, device=dev)
    dist.all_reduce(most, op=dist.ReduceOp.MAX, group=group)
    most = int(most.item())
    s_off = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
    r_off = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
    for r0 in range(0, most, chunk_rows):
        sc = [int(min(max(c - r0, 0), chunk_rows)) for c in send_counts]
        rc = [int(min(max(c - r0, 0), chunk_rows)) for c in recv_counts]
        if most <= chunk_rows:      # one round: no staging copies
            dist.all_to_all_single(recv, payload, output_split_sizes=rc, input_split_sizes=sc, group=group)
            break
        sbuf = torch.cat([payload[s_off[j] + r0: s_off[j] + r0 + sc[j]] for j in range(W)]) if sum(sc) else payload[:0]
        rbuf = torch.empty((sum(rc),) + tuple(payload.shape[1:]), dtype=payload.dtype, device=dev)
        dist.all_to_all_single(rbuf, sbuf.contiguous(), output_split_sizes=rc, input_split_sizes=sc, group=group)
        o = 0
        for j in range(W):
            recv[r_off[j] + r0: r_off[j] + r0 + rc[j]] = rbuf[o: o + rc[j]]
            o += rc[j]
    return recv


class ShardedRecruiter:
    def __init__(self, device_index=0, lib=None, torch_device=None, group=None, force_exchange=False):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group = group
        # force_exchange: run the exchange path (all-to-all, all-gathers, second engine) even with one rank — used by
        # the tests to exercise the RCCL / device-pointer plumbing on a single GPU
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        self.dev = torch.device(torch_device) if torch_device is not None else torch.device("cuda", device_index)
        self.local = Engine(device_index, lib)   # owns the read shard, the table, the local clouds
        self.glob = Engine(device_index, lib)    # owns the all-gathered clouds for the distance stage

    # results of the last run(): device resident (HBM) until read, like the inputs
    @property
    def unique_mask(self):
        """bool[n_rare]: rare k-mers selected by some edge (all ranks' edges when sharded)."""
        if "mask" not in self._host:
            self._host["mask"] = self._mask_t.cpu().numpy().astype(bool) if self._mask_t is not None else self.dist_engine.unique_mask()
        return self._host["mask"]

    @property
    def rare(self):
        """uint64[n_rare]: the rare k-mer set (2-bit codes, ascending)."""
        if "rare" not in self._host:
            self._host["rare"] = self._rare_t.cpu().numpy().astype(np.uint64) if self._rare_t is not None else self.dist_engine.kmers()
        return self._host["rare"]

    def close(self):
        self.local.close()
        self.glob.close()

    # ------------------------------------------------------------------ helpers
    def _sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def _all_gather_var(self, t, chunk_elems=None):
        """All-gather 1-D tensors of different lengths; returns (concatenated tensor, sizes list)."""
        if not dist.is_initialized():
            return t, [int(t.numel())]
        n = torch.tensor([t.numel()], dtype=torch.int64, device=self.dev)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        if chunk_elems is None:
            chunk_elems = max(1, CHUNK_BYTES // (t.element_size() * self.world))
        parts = [[] for _ in range(self.world)]
        for r0 in range(0, max(max(sizes), 1), chunk_elems):      # rounds of at most CHUNK_BYTES per rank (see CHUNK_BYTES)
            m = min(chunk_elems, max(max(sizes), 1) - r0)
            pad = torch.zeros(m, dtype=t.dtype, device=self.dev)
            mine = t[r0: r0 + m]
            pad[:mine.numel()] = mine
            out = [torch.empty_like(pad) for _ in range(self.world)]
            dist.all_gather(out, pad, group=self.group)
            for j in range(self.world):
                k = min(max(sizes[j] - r0, 0), m)
                if k:
                    parts[j].append(out[j][:k])
        flat = [p for ps in parts for p in ps]
        return (torch.cat(flat) if flat else t[:0]), sizes

    def load(self, packed, n_motif=1):
        self.local.load(packed, n_motif)
        up, _, _, _ = packed.units(n_motif)
        self._units_per_read = np.diff(up).astype(np.int64)

    # ------------------------------------------------------------------ one step
    def run(self, k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8, edge_cap=0):
        L, G, W, dev = self.local, self.glob, self.world, self.dev
        import time
        sec, t_last = {}, [time.perf_counter()]

        def lap(name):
            self._sync()
            now = time.perf_counter()
            sec[name] = round(sec.get(name, 0.0) + now - t_last[0], 4)
            t_last[0] = now
        self.sections = sec
        # 1. local counts
        L.count_kmers(k)
        st_local = L.stats()
        lap("count")
        if self.exchange:
            # 2. all-to-all of (key, pres | multi << 32) bucketed by owner
            n = L.table_size()
            keys = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
            pres = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            multi = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            self._sync()
            L.table_into(keys.data_ptr(), pres.data_ptr(), multi.data_ptr(), n)
            keys, pres, multi = keys[:n], pres[:n], multi[:n]
            lap("x_dump")
            own = _owner(keys, W)
            order = torch.sort(own.to(torch.uint8)).indices if W <= 256 else torch.argsort(own)     # one radix pass instead of eight
            send_counts = torch.bincount(own, minlength=W).to(torch.int64)
            payload = torch.stack([keys[order], (pres[order].to(torch.int64) & 0xFFFFFFFF) | (multi[order].to(torch.int64) << 32)], dim=1).contiguous()
            lap("x_bucket")
            recv_counts = torch.empty_like(send_counts)
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
            sc, rc = send_counts.tolist(), recv_counts.tolist()
            recv = all_to_all_rows(payload, sc, rc, group=self.group)
            lap("x_a2a")
            rk = recv[:, 0].contiguous()
            rp = (recv[:, 1] & 0xFFFFFFFF).to(torch.int32).contiguous()
            rm = (recv[:, 1] >> 32).to(torch.int32).contiguous()
            self._sync()
            lap("x_split")
            L.reset_table(k, max(int(rk.numel()), 1))
            lap("x_reset")
            if rk.numel():
                L.merge_table_ptr(rk.data_ptr(), rp.data_ptr(), rm.data_ptr(), rk.numel())
            self.exchange_bytes = int(payload.numel() * 8)
            lap("table_exchange")
        # 3. rare k-mers of the owned keys, gathered and sorted everywhere
        n_own = L.select_rare(max_nonuniq, lo, hi)
        st_owner = L.stats()
        if self.exchange:
            mine = torch.empty(max(n_own, 1), dtype=torch.int64, device=dev)
            self._sync()
            L.kmers_into(mine.data_ptr(), n_own)
            allr, _ = self._all_gather_var(mine[:n_own])
            rare = torch.sort(allr).values.contiguous()     # keys < 2^62: int64 order = unsigned order
            self._sync()
            L.set_kmers_ptr(rare.data_ptr(), rare.numel(), k)
            n_rare = int(rare.numel())
        else:
            n_rare = n_own
        lap("select")
        # 4. clouds of the local shard, gathered
        n_ce_local = L.build_clouds()
        lap("clouds")
        if self.exchange:
            U = L.n_units
            cp = torch.empty(U + 1, dtype=torch.int64, device=dev)
            ent = torch.empty(max(n_ce_local, 1), dtype=torch.int32, device=dev)
            self._sync()
            L.clouds_into(cp.data_ptr(), ent.data_ptr(), n_ce_local)
            sizes = (cp[1:] - cp[:-1]).contiguous()
            all_sizes, _ = self._all_gather_var(sizes)
            all_ent, _ = self._all_gather_var(ent[:n_ce_local])
            upr, _ = self._all_gather_var(torch.from_numpy(self._units_per_read).to(dev))
            cloud_ptr_g = torch.zeros(all_sizes.numel() + 1, dtype=torch.int64, device=dev)
            cloud_ptr_g[1:] = torch.cumsum(all_sizes, 0)
            unit_ptr_g = np.concatenate([[0], np.cumsum(upr.cpu().numpy())]).astype(np.int64)
            Rg, Ug = unit_ptr_g.size - 1, int(unit_ptr_g[-1])
            zeros_u = np.zeros(Ug, np.int64)
            G.load_arrays(np.zeros(0, np.uint8), np.zeros(Rg + 1, np.int64), unit_ptr_g, zeros_u, zeros_u)
            self._sync()
            G.set_kmers_ptr(rare.data_ptr(), n_rare, k)
            all_ent = all_ent.contiguous()
            G.set_clouds_ptr(cloud_ptr_g.data_ptr(), all_ent.data_ptr(), all_ent.numel())
            D = G
            lap("cloud_gather")
        else:
            D = L
        # 5. distance + filter on this rank's first k-mers
        D.reset_unique()
        n_edges = D.dist_edges(0, 2 ** 62, min_d, max_d, min_cov, rel_threshold, self.rank, W, edge_cap)
        st_d = D.stats()
        lap("dist")
        # 6. combine
        if self.exchange:
            mask = torch.empty(max(n_rare, 1), dtype=torch.uint8, device=dev)
            self._sync()
            D.unique_mask_into(mask.data_ptr())
            mask = mask[:n_rare]
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
            cnt = torch.tensor([n_edges, st_d["n_emissions"], st_local["n_bases"], st_local["n_windows"],
                                st_local["n_read_kmers"], st_owner["n_distinct"], st_owner["n_kept"], n_ce_local],
                               dtype=torch.int64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=self.group)
            c = cnt.tolist()
            n_unique = int(mask.sum().item())
            self._mask_t, self._rare_t, self._host = mask, rare, {}      # results stay on the device until asked for
        else:
            c = [n_edges, st_d["n_emissions"], st_local["n_bases"], st_local["n_windows"], st_local["n_read_kmers"],
                 st_owner["n_distinct"], st_owner["n_kept"], n_ce_local]
            n_unique = D.stats()["n_unique"]
            self._mask_t, self._rare_t, self._host = None, None, {}
        lap("combine")
        self.local_edges = n_edges
        self.dist_engine = D
        return dict(n_edges=c[0], n_emissions=c[1], n_bases=c[2], n_windows=c[3], n_read_kmers=c[4], n_distinct=c[5],
                    n_kept=c[6], n_cloud_entries=c[7], n_rare=n_rare, n_unique=n_unique,
                    local_emissions=st_d["n_emissions"], local_edges=n_edges, local_bases=st_local["n_bases"],
                    local_cloud_entries=n_ce_local, dist_kernel_ms=D.times()["dist_kernel_ms"],
                    n_spilled=st_d["n_spilled"], n_dist_passes=st_d["n_dist_passes"])
