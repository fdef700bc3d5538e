"""Seeded fixtures shared by the golden-capture script and the tests.

A fixture is (generator parameters, stage-2 parameters, stage-3 parameters).  The NCRF report
text is regenerated from the seed by the deterministic native generator (include/cfhost.h);
its SHA-256 is stored in the golden file so drift of the generator is detected.
"""
import hashlib
import os

FIXTURES = {
    # 200-bp unit, ~33 units per read: exercises large d, partial units, split records
    "tiny": dict(
        synth=dict(seed=11, unit_len=200, monomer_len=50, n_units=60, flank=60000, n_reads=30,
                   mean_len=6500, sigma=0.2, min_len=6000, max_len=7000, unit_div=0.03,
                   n_prefix=3, n_suffix=3, prefix_threshold=50000, p_split=0.15),
        stage2=dict(k=19, coverage=14, min_coverage=3, max_distance=150),
        stage3=dict(prefix_threshold=50000),
    ),
    # DXZ1-sized unit (2055 bp), ~5 units per read
    "hor2055": dict(
        synth=dict(seed=12, unit_len=2055, monomer_len=171, n_units=16, flank=70000, n_reads=40,
                   mean_len=9000, sigma=0.3, min_len=7000, max_len=12000, n_prefix=3, n_suffix=3,
                   prefix_threshold=50000, p_split=0.1),
        stage2=dict(k=19, coverage=12, min_coverage=3, max_distance=150),
        stage3=dict(prefix_threshold=50000),
    ),
    # low coverage + strict placer threshold: the placer runs out of qualifying reads and
    # writes a None tail after placing most reads
    "lowcov": dict(
        synth=dict(seed=13, unit_len=200, monomer_len=50, n_units=100, flank=60000, n_reads=20,
                   mean_len=6500, sigma=0.2, min_len=6000, max_len=7000, unit_div=0.015,
                   n_prefix=2, n_suffix=2, prefix_threshold=50000, p_split=0.1),
        stage2=dict(k=19, coverage=10, min_coverage=3, max_distance=150),
        stage3=dict(prefix_threshold=50000, min_inters=80),
    ),
}

# Fixtures outside the generic parity lists (their own tests): reads with symbols the 2-bit code does not have.
# "exotic": soft-masked (lower-case) stretches and N calls written into the alignment rows of the report after it was
# generated; max_distance 2 so that the host-emulated kernels can run the whole CLI against the reference's files.
EXTRA_FIXTURES = {
    "exotic": dict(
        synth=dict(seed=19, unit_len=200, monomer_len=50, n_units=60, flank=60000, n_reads=30,
                   mean_len=6500, sigma=0.2, min_len=6000, max_len=7000, unit_div=0.03,
                   n_prefix=3, n_suffix=3, prefix_threshold=50000, p_split=0.15),
        mutate=dict(seed=5, n_lower=18, lower_len=60, n_N=12),
        stage2=dict(k=19, coverage=14, min_coverage=3, max_distance=2),
        stage3=dict(prefix_threshold=50000),
    ),
    # "exotic_rare": the same base of the array turned into N in EVERY read row that holds it (n_shared loci): the windows over it
    # are k-mers with an N that are RARE — the reference selects them, puts them into clouds (read_kmer_cloud.py:25 upper-cases the
    # unit, the N stays) and writes them (distance_based_kmer_recruitment.py:47-53, :160-164); plus the random mutations of "exotic".
    "exotic_rare": dict(
        synth=dict(seed=23, unit_len=200, monomer_len=50, n_units=60, flank=60000, n_reads=30,
                   mean_len=6500, sigma=0.2, min_len=6000, max_len=7000, unit_div=0.03,
                   n_prefix=3, n_suffix=3, prefix_threshold=50000, p_split=0.15),
        mutate=dict(seed=7, n_lower=6, lower_len=60, n_N=4, n_shared=3, shared_lo=6, shared_hi=13),
        stage2=dict(k=19, coverage=14, min_coverage=3, max_distance=2),
        stage3=dict(prefix_threshold=50000),
    ),
}


def _spec(name):
    return FIXTURES[name] if name in FIXTURES else EXTRA_FIXTURES[name]


def mutate_report(path, seed, n_lower, lower_len, n_N, n_shared=0, shared_lo=0, shared_hi=0, shared_len=37):
    """Lower-case n_lower stretches of lower_len aligned bases and turn n_N bases into N, in the read rows of the report.
    n_shared: first, n_shared times, the middle base of a stretch of shared_len gap-free row symbols that occurs (once per row) in
    shared_lo .. shared_hi rows becomes N in every one of those rows."""
    import random
    rng = random.Random(seed)
    with open(path) as f:
        lines = f.read().split("\n")
    recs = [i for i, ln in enumerate(lines) if ln and not ln.startswith("#")][::2]
    for _ in range(n_shared):
        rows = {i: lines[i].split(None, 4)[4] for i in recs}
        count, twice = {}, set()
        for i, row in rows.items():
            mine = set()
            for c in range(len(row) - shared_len + 1):
                w = row[c:c + shared_len]
                if w.strip("ACGT"):
                    continue
                if w in mine:
                    twice.add(w)
                mine.add(w)
            for w in mine:
                count[w] = count.get(w, 0) + 1
        cand = sorted(w for w, n in count.items() if shared_lo <= n <= shared_hi and w not in twice)
        w = cand[rng.randrange(len(cand))]
        mut = w[:shared_len // 2] + "N" + w[shared_len // 2 + 1:]
        for i, row in rows.items():
            if w in row:
                head = lines[i].split(None, 4)
                lines[i] = " ".join(head[:4]) + " " + row.replace(w, mut)
    for what in ["lower"] * n_lower + ["N"] * n_N:
        i = rng.choice(recs)
        head = lines[i].split(None, 4)
        row = list(lines[i].split(None, 4)[4])
        c = rng.randrange(len(row))
        todo = lower_len if what == "lower" else 1
        while todo and c < len(row):
            if row[c] != "-":
                row[c] = row[c].lower() if what == "lower" else "N"
                todo -= 1
            c += 1
        lines[i] = " ".join(head[:4]) + " " + "".join(row)
    with open(path, "w") as f:
        f.write("\n".join(lines))


STAGE2_DEFAULTS = dict(k=19, min_coverage=4, min_nreads=0, max_nreads=2 ** 63 - 1, min_distance=1,
                       max_distance=150, bottom=0.9, top=3.0, kmer_survival_rate=0.34, max_nonuniq=3)
STAGE3_DEFAULTS = dict(n_motif=1, k_cloud=19, min_cloud_kmer_freq=2, min_kmer_mult=2, min_unit=2,
                       min_inters=10, prefix_threshold=50000)


def stage2_params(name):
    p = dict(STAGE2_DEFAULTS)
    p.update(_spec(name)["stage2"])
    return p


def stage3_params(name):
    p = dict(STAGE3_DEFAULTS)
    p.update(_spec(name)["stage3"])
    return p


def make_report(name, outdir):
    """Write the fixture's NCRF report into outdir; returns its path."""
    from centroflye_amd import _host
    spec = _spec(name)
    tag = hashlib.sha1(repr(sorted(spec["synth"].items()) + sorted(spec.get("mutate", {}).items())).encode()).hexdigest()[:10]
    path = os.path.join(outdir, f"{name}_{tag}.ncrf")
    if not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        _host.synth(report_path=tmp, pack=False, **spec["synth"])
        if "mutate" in spec:
            mutate_report(tmp, **spec["mutate"])
        os.replace(tmp, path)
    return path


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()
