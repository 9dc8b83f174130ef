"""GPU: bench.py's N > 1 path with bags in flight -- two (the default) or three HIP streams per rank, ONE all-reduce per bag issued on the bag's
stream (pipeline.BagsInFlight.all_reduce_slot) -- rehearsed with two ranks that share the box's one GPU (gloo instead of
RCCL, which refuses two ranks on one device; MMF_BENCH_REHEARSAL=1).  Every rank must issue exactly the same number of
collectives (a mismatch is a hang on RCCL) and the run must print one well-formed JSON line with n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("inflight", [3, 2, 1])
def test_two_rank_rehearsal_counts_collectives(tmp_path, inflight):
    steps, warmup, blocks = 4, 2, 2
    tag = str(tmp_path / "collectives")
    env = dict(os.environ, MMF_BENCH_REHEARSAL="1", MMF_BENCH_COUNT_COLLECTIVES=tag, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps),
           "--warmup", str(warmup), "--blocks", str(blocks), "--bag", "3000", "--inflight", str(inflight), "--no-extras",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["warmup"] == warmup and out["blocks"]["count"] == blocks
    assert out["scaling"] == "weak" and out["value"] > 0
    counts = [json.load(open(f"{tag}.rank{k}")) for k in range(2)]
    assert counts[0] == counts[1], counts                       # equal on every rank, or RCCL would hang
    # one all-reduce per bag (warm-up + timed blocks) + one MAX all-reduce of the block time per block
    assert counts[0]["all_reduce"] == warmup + steps * blocks + blocks, counts
