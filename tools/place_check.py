#!/usr/bin/env python3
"""Developer tool (GPU box): who is right where the GPU placement and the C placer first differ?  Recomputes the scores of
the two reads in question from first principles (reference cloud_contig.py:26-41, :87-95, read_placer.py:54-78): the
events a (k-mer, position) pair raises = one seed event if the k-mer was frequent somewhere and present at the position
when the stage began + one when its count reaches the threshold during the stage; score[r][off] = events over the read's
(unit, k-mer) pairs.  usage: tools/place_check.py <reads> <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine
from oracle import cport

n, seed = int(sys.argv[1]), int(sys.argv[2])
force = int(sys.argv[3]) if len(sys.argv) > 3 else -1      # self-check: recompute the scores of the pick at this output index even when both agree
lib = None
if os.environ.get("CF_LIB"):
    from centroflye_amd import _lib
    lib = _lib.load(os.environ["CF_LIB"])
THR = 2
pk = _host.synth(seed=seed, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
up, _, _, _ = pk.units(1)
up = np.asarray(up)
cls = pk.classify(50000)
rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
e = Engine(0, lib)
e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds(); e.reset_unique()
e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
gk = e.kmers()[e.unique_mask()]
e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(2)
cp, ent = e.clouds()
got = e.place_reads(cls, rank, 2, 2, 10, 3)
e.close()
want = cport.place_reads(cls, rank, up, cp, ent, gk.size, 2, 2, 10, 3)
norm = lambda rows: [(r, p, a, 0 if a < 0 else b) for r, p, a, b in rows]      # (prefix / None rows carry no scores)
g = norm(zip(*[x.tolist() for x in got])); w = norm(zip(*[x.tolist() for x in want]))
print("classes", np.bincount(cls, minlength=3).tolist(), flush=True)
diff = [i for i, (a, b) in enumerate(zip(g, w)) if a != b]
if not diff and force < 0:
    print("identical"); sys.exit(0)
t = diff[0] if diff else force
print("first difference at output index", t, "GPU", g[t], "oracle", w[t], flush=True)
n_prefix = int((cls == 0).sum())
# postings over all reads: k-mer -> (read, unit index)
unit_read = np.repeat(np.arange(n, dtype=np.int64), np.diff(up))
ent_unit = np.repeat(np.arange(cp.size - 1, dtype=np.int64), np.diff(cp))
order = np.argsort(ent, kind="stable")
post_unit = ent_unit[order]
pptr = np.zeros(gk.size + 1, np.int64); np.cumsum(np.bincount(ent, minlength=gk.size), out=pptr[1:])
pos = np.full(n, -1, np.int64); when = np.full(n, 2 ** 40, np.int64)
for i in range(t):
    assert g[i] == w[i]
    r, p = g[i][0], g[i][1]
    pos[r], when[r] = p, i
def score(r, off):
    s0 = s1 = 0
    for u in range(up[r], up[r + 1]):
        i = u - up[r]; q = off + i; hits = 0
        for x in ent[cp[u]:cp[u + 1]].tolist():
            pu = post_unit[pptr[x]:pptr[x + 1]]
            pr = unit_read[pu]; pi = pu - up[pr]
            placed = pos[pr] >= 0
            at_q = placed & (pos[pr] + pi == q)
            pre_q = int((at_q & (when[pr] < n_prefix)).sum()); all_q = int(at_q.sum())
            # frequent somewhere when the stage began: some position holds >= THR prefix reads with x
            pre = placed & (when[pr] < n_prefix)
            freq_pre = False
            if pre.sum() >= THR:
                _, c = np.unique(pos[pr][pre] + pi[pre], return_counts=True)
                freq_pre = bool((c >= THR).any())
            ev = (1 if (freq_pre and pre_q >= 1) else 0) + (1 if (pre_q < THR <= all_q) else 0)
            hits += ev
        s1 += hits; s0 += hits > 0
    return s0, s1
for name, row in (("GPU pick", g[t]), ("oracle pick", w[t])):
    r, off = row[0], row[1]
    print(name, pk.ids[r], "rank", int(rank[r]), "class", int(cls[r]), "claimed", row[1:], "recomputed (s0, s1) at that offset:", score(r, off), flush=True)
# the other side's view of each read: where does each implementation place the other one's pick later?
for name, rows, other in (("GPU", g, w[t][0]), ("oracle", w, g[t][0])):
    j = [i for i, x in enumerate(rows) if x[0] == other][0]
    print(f"{name} places {pk.ids[other]} at output index {j}: {rows[j][1:]}")
