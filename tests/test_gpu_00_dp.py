"""Data-parallel step end to end with two ranks sharing the one GPU of the test box (gloo carries the collectives: RCCL
wants one device per rank): both wire formats must leave the replicas bit-identical after training steps that include
density-grid refreshes, graph replays and the timed (eager) variants of the step."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("wire,mode", [("f32", "shard"), ("bf16", "shard"), ("f32", "allreduce"), ("bf16", "allreduce")])
def test_two_ranks_stay_in_sync(wire, mode):
    env = dict(os.environ, NGP_DIST_BACKEND="gloo", NGP_LOCAL_DEVICE="0", MASTER_ADDR="127.0.0.1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    port = 29600 + (os.getpid() % 200) + (0 if wire == "bf16" else 1) + (0 if mode == "shard" else 2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12",
           "--warmup", "4", "--burnin", "20", "--psnr-iters", "40", "--views", "6", "--res", "96", "--rays", "1024",
           "--no-cpu-baseline", "--no-secondary", "--grad-wire", wire, "--dp-mode", mode]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    assert res["config"]["parallelism"] == "dp2" and res["config"]["grad_wire"] == wire
    assert res["config"]["dp_mode"] == mode and res["config"]["ranks_seen"] == 2
    if mode == "shard":
        assert res["config"]["collective_ms_per_step"] > 0
    assert res["config"]["replicas_in_sync"] is True
    assert res["value"] > 0 and res["psnr"]["value"] > 5.0
