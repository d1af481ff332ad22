"""bench.py's multi-GPU code paths on the one-GPU box:
  * `--force-dist`: the step goes through libmfsr_dist.so (mfsr_dist_process_burst, one-rank RCCL communicator) -- the code the
    driver's N > 1 runs execute (unique-id handling, frame pointer table, status probe, launch timing through the dist
    context's burst);
  * `--gpus 2` with MFSR_DIST_BACKEND=gloo: two ranks sharing the GPU, torch.distributed mirror, stripes exchange staged
    through the host -- a functional rehearsal of the strong-scaling schedule, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


@pytest.mark.parametrize("exchange", ["stripes", "reduce_scatter"])
def test_bench_force_dist_single_rank(exchange):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--exchange", exchange, "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--workload", "1080p5_gray_x2"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["launches_timed"] > 0
    assert d["scaling"] == "strong"


def test_bench_two_ranks_gloo_rehearsal():
    env = dict(os.environ, MFSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "1080p5_gray_x2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["burst_frames"] == 5
    assert "stripes" in d["config"]["parallelism"] and "rehearsal" in d["config"]
