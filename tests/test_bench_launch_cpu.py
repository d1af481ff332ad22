"""`python3 bench.py --gpus N` must be startable as a plain command line (the driver's scaling command): the parent
spawns N fresh rank processes before anything touches a GPU, relays rank 0's JSON line and fails if any rank fails.
Exercised here with the `--dry-run-ranks` hook (ranks print their launch environment and exit; no GPU needed)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=120)


def test_plain_command_line_spawns_one_rank_per_gpu():
    p = _run(["--gpus", "4", "--steps", "1", "--warmup", "0", "--dry-run-ranks"])
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                      # rank 0's line only
    d = json.loads(lines[0])
    assert d["rank"] == 0 and d["world"] == 4 and d["gpus"] == 4 and d["child"] is True
    assert d["master"].startswith("127.0.0.1:")


def test_a_failing_rank_fails_the_run():
    p = _run(["--gpus", "3", "--dry-run-ranks"], env={"MFSR_BENCH_FAIL_RANK": "2"})
    assert p.returncode != 0
    assert "rank 2 exited with code 3" in p.stderr


def test_launcher_environment_is_respected():
    """under torch.distributed.run (WORLD_SIZE set) the script is a rank, not a parent"""
    p = _run(["--gpus", "2", "--dry-run-ranks"], env={"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1",
                                                     "MASTER_PORT": "29999"})
    assert p.returncode == 0
    d = json.loads(p.stdout.strip())
    assert d["rank"] == 1 and d["world"] == 2 and d["child"] is False
