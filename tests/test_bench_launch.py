"""`python bench.py --gpus N` as a bare command starts its own rank processes: the launch plumbing is
exercised here without a GPU through --dry-launch (the ranks join a gloo group and report what they were given)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, gpus=2):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-launch"],
                          env=env, capture_output=True, text=True, timeout=300)


def test_bare_command_starts_one_process_per_gpu():
    p = _run()
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                      # stdout carries the one JSON line and nothing else
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == 2 and out["group_size"] == 2
    ranks = sorted(out["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1]
    assert all(r["world_size_env"] == 2 for r in ranks)
    assert [r["local_rank"] for r in ranks] == [0, 1]
    assert len({r["pid"] for r in ranks}) == 2 and os.getpid() not in {r["pid"] for r in ranks}


def test_a_failed_rank_fails_the_command():
    p = _run({"BENCH_DRY_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]


def test_parent_never_touches_the_gpu_stack():
    """The launching parent must not import torch or load the HIP library before the children exist."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def launch_ranks")]
    body = src[src.index("def launch_ranks"):src.index("def dry_launch")]
    top_level_imports = [ln for ln in src.splitlines() if ln.startswith(("import ", "from "))]
    assert not any("torch" in ln or "ml_amd" in ln or "numpy" in ln for ln in top_level_imports)
    assert "import torch" not in body and "ml_amd" not in body and head is not None
