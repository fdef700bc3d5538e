"""The N > 1 path (reads sharded, all-to-all count merge, all-gathers, first-k-mer partition) on
world_size 2 with gloo, CPU tensors and the host-emulated kernels; compared with the C oracle run
on the union of the shards."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_2_equals_single_process_oracle(emu_lib):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "sharded_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("SHARDED_RESULT ")]
    assert line, p.stdout[-2000:]
    res = json.loads(line[0].split(" ", 1)[1])
    assert res["world"] == 2 and res["n_rare"] > 100 and res["n_edges"] > 100
    assert res["rare"] and res["unique"] and res["counters"] and res["edge_checksum"], res
