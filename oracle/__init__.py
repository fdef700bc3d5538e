"""oracle/ — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in ``centroflye_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, and only as
the checker / the timed CPU baseline — never as the thing shipped.

Parity status: PINNED.  Every function here is checked against golden vectors captured by
importing the reference itself (``/root/reference/scripts``) in the build container with
``tests/golden/make_golden.py``; see ``tests/test_oracle_golden.py``.

Modules
  ncrf.py     A0   scripts/ncrf_parser.py:28-59, :61-118, :120-145
  recruit.py  A1-A7 scripts/distance_based_kmer_recruitment.py:39-171, scripts/read_kmer_cloud.py:17-40
  placer.py   A4, A8-A9 scripts/read_kmer_cloud.py:43-54, scripts/cloud_contig.py:9-41,:87-95,
              scripts/read_placer.py:35-128
  c/          the same A1-A6 in plain C for the timed CPU baseline
"""
