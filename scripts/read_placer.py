#!/usr/bin/env python3
"""Entry point with the reference's script name and CLI (centroFlye.py:172-184 / :196-204 call
`python -u scripts/read_placer.py ...`); the implementation lives in centroflye_amd/read_placer.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centroflye_amd.read_placer import *  # noqa: E402,F401,F403
from centroflye_amd.read_placer import main  # noqa: E402

if __name__ == "__main__":
    main()
