#!/usr/bin/env python3
"""Entry point with the reference's program name and arguments (scripts/read_recruitment/run_read_recruitment.sh:42 runs
`rr unit.fasta reads.fasta out.fasta threshold` per input split); implementation: centroflye_amd/read_recruitment.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centroflye_amd.read_recruitment import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
