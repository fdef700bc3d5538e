#!/usr/bin/env python3
"""Developer tool (BUILD CONTAINER ONLY: needs /root/reference, no GPU): randomised differential test of the importable API of SURVEY §8(b)
that no command line reaches (row A10): centroflye_amd.cloud_contig against the reference's cloud_contig — CloudContig.add_read (returned
events in order, every attribute: max_pos, clouds, freq_clouds, freq_kmers, kmer_positions, read_positions, coverage), update_mapping_scores
(fresh and continued), calc_inters_score (windows, thresholds), get_spread_kmers, map_reads, map_reads_fast — on random clouds.
usage: tools/fuzz_api_vs_reference.py [cases] [--seed S]"""
import importlib.util, json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/scripts"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class Cloud:      # the attributes both implementations read (read_kmer_cloud.ReadKMerCloud: r_id, kmers = list of sets, all_kmers)
    def __init__(self, r_id, kmers):
        self.r_id, self.kmers = r_id, kmers
        self.all_kmers = set().union(*kmers) if kmers else set()


def plain(x):
    if isinstance(x, dict):
        return {k: plain(v) for k, v in x.items()}
    if isinstance(x, (set, frozenset)):
        return sorted(x)
    return x


def state(cc):
    return dict(max_pos=cc.max_pos, clouds={p: dict(c) for p, c in cc.clouds.items()}, freq_clouds={p: sorted(s) for p, s in cc.freq_clouds.items() if s},
                freq_kmers=sorted(cc.freq_kmers), kmer_positions={k: sorted(v) for k, v in cc.kmer_positions.items()}, read_positions=dict(cc.read_positions),
                coverage={p: c for p, c in cc.coverage.items() if c})


def scores_plain(sc):
    return {r: {o: dict(c) for o, c in by.items() if c} for r, by in sc.items() if any(by.values())}


def main():
    if not os.path.isdir(REF):
        sys.exit("this tool needs /root/reference (build container only)")
    sys.dont_write_bytecode = True
    R = load(os.path.join(REF, "cloud_contig.py"), "ref_cloud_contig")
    from centroflye_amd import cloud_contig as O
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 1
    rng = random.Random(seed)
    bad = 0
    for i in range(n_cases):
        alphabet = [f"K{j}" for j in range(rng.choice([5, 30, 200]))]
        freq = rng.choice([0, 1, 2, 3])
        rc, oc = R.CloudContig(freq), O.CloudContig(freq)
        reads = {}
        for r in range(rng.choice([1, 4, 12])):
            reads[f"r{r}"] = Cloud(f"r{r}", [set(rng.sample(alphabet, rng.randint(0, min(len(alphabet), 8)))) for _ in range(rng.randint(0, 7))])
        diffs = []
        r_sc = o_sc = None
        k2p = {}
        for r_id, cl in reads.items():
            for u, c in enumerate(cl.kmers):
                for x in c:
                    k2p.setdefault(x, []).append((r_id, u))
        for r_id, cl in reads.items():
            if rng.random() < 0.8:
                pos = rng.randint(0, 6)
                ev_r, ev_o = rc.add_read(cl, pos), oc.add_read(cl, pos)
                if sorted(ev_r) != sorted(ev_o) or len(ev_r) != len(ev_o):
                    diffs.append("add_read events")
                if state(rc) != state(oc):
                    diffs.append("state after add_read")
                keep = rng.random() < 0.7      # (continue the scores of the earlier reads, or start over: both sides alike)
                r_sc = R.update_mapping_scores(rc, k2p, ev_r, r_sc if keep else None)
                o_sc = O.update_mapping_scores(oc, k2p, ev_r, o_sc if keep else None)
                if scores_plain(r_sc) != scores_plain(o_sc):
                    diffs.append("update_mapping_scores")
        # continued accumulation over all events of the contig, both sides the same way
        allev = sorted((k, p) for k in rc.freq_kmers for p in rc.kmer_positions[k])
        a = R.update_mapping_scores(rc, k2p, allev[:len(allev) // 2], None); a = R.update_mapping_scores(rc, k2p, allev[len(allev) // 2:], a)
        b = O.update_mapping_scores(oc, k2p, allev[:len(allev) // 2], None); b = O.update_mapping_scores(oc, k2p, allev[len(allev) // 2:], b)
        if scores_plain(a) != scores_plain(b):
            diffs.append("update_mapping_scores (continued)")
        for cl in reads.values():
            kw = dict(min_position=rng.choice([0, 0, 2]), max_position=rng.choice([None, None, 3, rc.max_pos]), min_unit=rng.choice([0, 1, 2]), min_inters=rng.choice([0, 1, 3, 10]))
            if kw["max_position"] is not None and kw["max_position"] > rc.max_pos:
                kw["max_position"] = rc.max_pos
            try:
                want = rc.calc_inters_score(cl, **kw)
            except AssertionError:
                want = "assert"
            try:
                got = oc.calc_inters_score(cl, **kw)
            except AssertionError:
                got = "assert"
            if (want if want == "assert" else (tuple(want[0]), want[1])) != (got if got == "assert" else (tuple(got[0]), got[1])):
                diffs.append(f"calc_inters_score {kw}")
        n = rng.choice([0, 1, 5])
        if rc.get_spread_kmers(n) != oc.get_spread_kmers(n):
            diffs.append("get_spread_kmers")
        thr = (rng.choice([0, 1, 2, 5]), rng.choice([0, 1, 3, 10]))
        for fn in ("map_reads", "map_reads_fast"):
            try:
                w = getattr(R, fn)(rc, reads, threshold=thr)
            except (AssertionError, TypeError) as ex:
                w = type(ex).__name__
            try:
                g = getattr(O, fn)(oc, reads, threshold=thr)
            except (AssertionError, TypeError) as ex:
                g = type(ex).__name__
            if isinstance(w, str) or isinstance(g, str):
                if w != g:
                    diffs.append(f"{fn}: {w if isinstance(w, str) else 'ok'} vs {g if isinstance(g, str) else 'ok'}")
            elif fn == "map_reads":
                if (dict(w[0]), {k: tuple(v) for k, v in w[1].items()}) != (dict(g[0]), {k: tuple(v) for k, v in g[1].items()}):
                    diffs.append(fn)
            elif dict(w[0]) != dict(g[0]) or scores_plain(w[1]) != scores_plain(g[1]):
                diffs.append(fn)
        if diffs:
            bad += 1
            print("DIFFERENCE:", json.dumps(dict(case=i, freq=freq, differences=sorted(set(diffs))[:8])), flush=True)
    print(json.dumps(dict(seed=seed, cases=n_cases, different=bad)))
    sys.exit(1 if bad else 0)


main()
