"""Read recruitment, the stage before the hot path.

Mirror of the reference's ``scripts/read_recruitment/rr.cpp`` (a C++ program around vendored edlib, run once per input
split by ``run_read_recruitment.sh:42``): keep every read within ``threshold`` edits of the HOR unit or of its reverse
complement, the unit being allowed to sit anywhere inside the read (edlib mode HW).  Same command line

    rr.py unit.fasta reads.fasta[.gz] output.fasta edit_distance_threshold

and the same output (``>name`` + sequence on one line, input order, rr.cpp:83-87).  The distances are computed on the
GPU (``cf_rr_distances``, one wavefront per read and strand); reading the FASTA / FASTQ input (kseq semantics: name =
header up to the first white space, multi-line sequences joined, qualities ignored) stays on the host.
One deliberate difference: rr.cpp builds the reverse complement into a buffer without a terminating NUL and reverses it
with strlen (rr.cpp:56-62, undefined behaviour that happens to work when the byte after the buffer is zero); here the
reverse complement is simply correct.
"""
import gzip
import sys

import numpy as np

BATCH_BASES = 1 << 30


def read_first_seq(path):
    for name, seq in iter_seqs(path):
        return seq
    raise ValueError(f"{path}: no sequence")


def iter_seqs(path):
    """(name, sequence bytes) of a FASTA or FASTQ file, plain or gzip — what kseq_read yields (rr.cpp:67-73):
    name = header up to the first white space; sequence lines joined until a line that starts with '>', '@' or '+';
    after '+' as many quality characters as the sequence has bases are skipped."""
    with open(path, "rb") as probe:
        gz = probe.read(2) == b"\x1f\x8b"
    f = gzip.open(path, "rb") if gz else open(path, "rb")
    with f:
        lines = iter(f)
        header = next((ln for ln in lines if ln[:1] in (b">", b"@")), None)
        while header is not None:
            fields = header[1:].split(None, 1)
            name = fields[0] if fields else b""
            parts, nxt, plus = [], None, False
            for ln in lines:
                c = ln[:1]
                if c in (b">", b"@"):
                    nxt = ln
                    break
                if c == b"+":
                    plus = True
                    break
                ln = ln.rstrip(b"\r\n")
                if ln:
                    parts.append(ln)
            seq = b"".join(parts)
            if plus:
                got = 0
                for ln in lines:
                    got += len(ln.rstrip(b"\r\n"))
                    if got >= len(seq):
                        break
                nxt = next((ln for ln in lines if ln[:1] in (b">", b"@")), None)
            yield name, seq
            header = nxt


def recruit(unit, reads_path, output_path, threshold, engine=None):
    """Writes the recruited reads; returns (reads seen, reads recruited)."""
    from .engine import Engine
    own = engine is None
    engine = engine or Engine(0)
    n_seen = n_kept = 0
    try:
        with open(output_path, "wb") as out:
            batch, size = [], 0

            def flush():
                nonlocal n_kept, batch, size
                if not batch:
                    return
                off = np.zeros(len(batch) + 1, np.int64)
                np.cumsum([len(s) for _, s in batch], out=off[1:])
                flat = np.frombuffer(b"".join(s for _, s in batch), dtype=np.uint8)
                fwd, rc = engine.rr_distances(unit, flat, off, threshold)
                for (name, seq), a, b in zip(batch, fwd, rc):
                    if a != -1 or b != -1:      # rr.cpp:84
                        out.write(b">" + name + b"\n" + seq + b"\n")
                        n_kept += 1
                batch, size = [], 0

            for name, seq in iter_seqs(reads_path):
                batch.append((name, seq))
                size += len(seq)
                n_seen += 1
                if size >= BATCH_BASES:
                    flush()
            flush()
    finally:
        if own:
            engine.close()
    return n_seen, n_kept


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) != 5 or argv[1] == "-h":
        print("Usage: rr.py unit.fasta reads.fasta.gz output.fasta edit_distance_threshold", end="")
        return 0
    unit = read_first_seq(argv[1])
    recruit(unit, argv[2], argv[3], int(argv[4]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
