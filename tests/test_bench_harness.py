"""bench.py's own plumbing, checked on CPU with the host-emulated library (its `--lib` test hook): the JSON contract,
N > 1 both ways the brief allows it to start (launched per rank by torch.distributed.run, or spawning its own ranks),
the rendezvous file, the rank-max timing through cf_comm_allreduce_i64 — and that nothing here needs torch."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libcfhip_emu.so")
ARGS = ["--lib", EMU, "--reads", "6", "--steps", "1", "--warmup", "0", "--edge-cap", "1000", "--transfer-steps", "1", "--no-cpu-baseline", "--steps-b", "1", "--steps-c", "1", "--reads-c", "3", "--units-c", "24"]


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-3000:]
    return json.loads(lines[0])


def _check(res, n):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "value_incl_transfers"):
        assert key in res, key
    assert res["n_gpus"] == n and res["steps"] == 1 and res["scaling"] == "weak" and res["vs_baseline"] is None and res["value"] > 0
    assert "workload" in res["config"] and "var_len" in res["config"]["workload"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "whole_step_frac")) <= set(res["roofline"]) and res["roofline"]["traffic"] is None
    assert res["steps_identical"] and res["counters"]["n_bases"] > 100000 * n
    assert set(("bound_by_contract", "lds", "l2_hit", "hbm_side_gbps")) <= set(res["roofline"]) and res["roofline"]["bound_by_contract"] == "hbm"
    wb = res["workload_b"]      # the same reads with point substitutions (var_len 1), its own timed steps
    assert wb["steps"] == 1 and wb["value"] > 0 and wb["steps_identical"] and "var_len 1" in wb["workload"] and wb["counters"]["n_bases"] > 100000 * n
    assert set(("count", "select", "clouds", "postings", "dist")) <= set(wb["stage_ms_per_step"]) and wb["parity_vs_committed_oracle"] is None
    wc = res["workload_c"]      # cenX-shaped reads (BASELINE configs[4]'s regime), N = 1 only
    if n == 1:
        assert set(wc) == {"var_len_8", "var_len_1"}
        for w in wc.values():
            assert w["steps"] == 1 and w["value"] > 0 and w["steps_identical"] and "cenX-shaped" in w["workload"] and w["counters"]["n_bases"] > 50000
            assert set(("kernel_ms", "pair_emissions_per_s", "frac", "algorithmic_bytes_per_launch")) <= set(w["roofline"]) and w["parity_vs_committed_oracle"] is None
    else:
        assert wc is None


def test_single_process(emu_lib, tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    _check(_line(p.stdout), 1)


def test_spawns_its_own_ranks(emu_lib, tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = _line(p.stdout)
    _check(res, 2)
    assert res["exchange_bytes_per_step"] > 0


def test_eight_ranks_more_ranks_than_some_have_reads(emu_lib, tmp_path):
    """World 8 before an 8-GPU node runs it: `python bench.py --gpus 8` on the emulated kernels with the file transport —
    eight ranks meet in every exchange (table all-to-all, rare-list and cloud gathers, unique-mask reduction, counters), with 2 reads per
    rank; the line carries the rank-local roofline figures and, without --no-cpu-baseline, the per-GPU cpu_baseline."""
    args = [x for x in ARGS if x != "--no-cpu-baseline"] + ["--cpu-parts", "2"]
    args[args.index("--reads") + 1] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + args, capture_output=True, text=True, timeout=1500, cwd=tmp_path,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = _line(p.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "ms_per_step", "scaling", "roofline", "cpu_baseline", "config"):
        assert key in res, key
    assert res["n_gpus"] == 8 and res["value"] > 0 and res["scaling"] == "weak" and res["steps_identical"] and res["exchange_bytes_per_step"] > 0
    assert res["counters"]["n_bases"] > 8 * 2 * 5000 and "x8" in res["config"]["parallelism"]
    assert res["roofline"]["algorithmic_bytes_per_launch"] > 0 and res["roofline"]["kernel"] == "cf_dist_kernel"
    assert res["cpu_baseline"]["kind"] == "port" and res["cpu_baseline"]["value"] > 0 and "N = 1 configuration" in res["cpu_baseline"]["sample"]
    assert res["workload_b"]["value"] > 0 and res["workload_c"] is None


def test_two_ranks_carry_a_cpu_baseline(emu_lib, tmp_path):
    """N > 1 lines have a cpu_baseline too: rank 0 times the per-GPU workload of the weak-scaling line on its host
    after the last collective; the other ranks wait in the closing barrier."""
    args = [x for x in ARGS if x != "--no-cpu-baseline"] + ["--cpu-parts", "2"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True, timeout=600, cwd=tmp_path,
                       env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = _line(p.stdout)
    _check(res, 2)
    cb = res["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["extrapolated"] and "N = 1 configuration" in cb["sample"]


def test_under_torch_distributed_run(emu_lib, tmp_path):
    pytest.importorskip("torch")        # only the LAUNCHER is torch's; bench.py itself imports none of it
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp_path, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    _check(_line(p.stdout), 2)


def test_package_and_bench_are_torch_free():
    """north_star: host code calls the HIP kernels through a thin ctypes layer, no PyTorch."""
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for d in ("centroflye_amd", "scripts"):
        for dirpath, _, names in os.walk(os.path.join(ROOT, d)):
            files += [os.path.join(dirpath, n) for n in names if n.endswith(".py")]
    for fn in files:
        with open(fn) as f:
            assert not re.search(r"^\s*(import|from)\s+torch\b", f.read(), flags=re.M), fn
