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


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (no WORLD_SIZE): bench.py starts the two ranks itself, as
    children, and relays rank 0's line -- n_gpus must be 2, and every rank must have issued the same collectives."""
    steps, warmup, blocks = 3, 1, 2
    tag = str(tmp_path / "collectives")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MMF_BENCH_REHEARSAL="1", MMF_BENCH_COUNT_COLLECTIVES=tag, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup),
           "--blocks", str(blocks), "--bag", "3000", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["config"]["bags_in_flight"] == 2
    counts = [json.load(open(f"{tag}.rank{k}")) for k in range(2)]
    assert counts[0] == counts[1] and counts[0]["all_reduce"] == warmup + steps * blocks + blocks, counts


def test_gpus_flag_must_match_the_world_size():
    """A launcher that starts a different number of ranks than --gpus says is an error, not a line with another n_gpus."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-extras", "--no-cpu-baseline", "--bag", "1000"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
