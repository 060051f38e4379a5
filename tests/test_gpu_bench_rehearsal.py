"""`python bench.py --gpus N` end to end with N > 1 rank processes on the ONE GPU of the test box: self-launch, process group,
row sharding of the chunk-wise synthetic data, the statistics all-reduce inside mlhip_em_iterate / mlhip_kmeans_step, the
max-over-ranks timing and the single JSON line. The transport is gloo on the host (`--allreduce gloo`): RCCL refuses two ranks
on one device, so the NCCL call itself is the only piece this cannot exercise (tests/test_gpu_rccl.py runs it with one rank).
The sharded job must reproduce the single-process numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_two_rank_em_line_matches_the_single_rank_run():
    one = _bench("--samples", "400000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    two = _bench("--gpus", "2", "--allreduce", "gloo", "--n", "400000", "--steps", "3", "--warmup", "1")
    # two ranks on the ONE GPU of this box: `n_gpus` counts physical GPUs, `units` the ranks; the line says it is a rehearsal
    assert two["n_gpus"] == 1 and two["units"] == 2 and "rehearsal" in two
    assert two["n_local"] == [200000, 200000] and two["steps"] == 3
    a, b = one["config"]["final_mean_log_likelihood"], two["config"]["final_mean_log_likelihood"]
    assert abs(a - b) <= 1e-12 * abs(a)
    assert "roofline" in two and two["roofline"]["kernel_ms"]["em_estep"] > 0
    # the fields a first multi-GPU run is diagnosed with: every line has them, a single rank reports no all-reduce time
    for line in (one, two):
        assert line["ms_per_step_min"] <= line["ms_per_step_max"] <= line["ms_per_step"] * 1.5 + 1.0
        assert len(line["allreduce_ms_per_rank"]) == line["units"]
    assert one["allreduce_ms"] == 0
    assert one["roofline"]["exp_runs_in"].startswith("em_mstats")        # d = 32, K = 64: self-normalising statistics kernel


def test_three_rank_kmeans_and_diag_lines():
    one = _bench("--workload", "kmeans", "--samples", "600000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    three = _bench("--gpus", "3", "--allreduce", "gloo", "--workload", "kmeans", "--samples", "600000", "--steps", "3", "--warmup", "1")
    assert three["n_local"] == [200000, 200000, 200000]
    assert abs(one["config"]["inertia"] - three["config"]["inertia"]) <= 1e-12 * one["config"]["inertia"]
    d1 = _bench("--workload", "em-diag", "--samples", "300000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    d3 = _bench("--gpus", "3", "--allreduce", "gloo", "--workload", "em-diag", "--samples", "300000", "--steps", "3", "--warmup", "1")
    a, b = d1["config"]["final_mean_log_likelihood"], d3["config"]["final_mean_log_likelihood"]
    assert abs(a - b) <= 1e-12 * abs(a)
