"""bench.py must start its own ranks when it is called as a plain command (`python bench.py --gpus N`, the form the driver's
scaling run uses): without WORLD_SIZE in the environment it re-executes itself under torch.distributed.run.  The rank plumbing
(process group, barriers, MAX over ranks, gather, one JSON line from rank 0) is exercised here on gloo with a stub body."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


def test_plain_command_spawns_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    # rank r accumulated 3 * (r + 1) * [0, 1, 2, 3]
    assert out["gathered"] == [[0.0, 3.0, 6.0, 9.0], [0.0, 6.0, 12.0, 18.0]]


def test_rank_count_mismatch_is_an_error_not_an_assert():
    r = _run(["--gpus", "2", "--launcher-selftest"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)
