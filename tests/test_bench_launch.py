"""bench.py's own launch plumbing on CPU: `python bench.py --gpus 2 --dry-run` must start two ranks by itself (gloo), shard the
chunk list over them, gather, and report n_gpus == 2; a launcher / --gpus mismatch must be loud."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_2_dry_run_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["config"]["gathered_ok"] is True
    assert out["value"] > 0 and out["scaling"] == "weak"


def test_single_rank_dry_run():
    r = _run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1


def test_launcher_mismatch_is_loud():
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "n_gpus would be misreported" in (r.stderr + r.stdout)
