"""bench.py's N > 1 launcher on the CPU: `python bench.py --gpus N` starts N FRESH child processes (this parent never
touches a GPU and never re-executes itself), they form a gloo process group, cut the 4K frame into tile columns, work out the per-frame
exchange plan and reduce over the ranks; rank 0 prints ONE JSON line.  `--workload launcher_dry_run` runs exactly that and nothing else (no
device call, no oracle, nothing measured).  Two injected failures must end the whole job with a non-zero exit instead of a hang: a rank that
dies, and a communicator that holds fewer ranks than the job (what the assert on aomhip_comm_info guards in the real run)."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT


def run(n, extra_env=None, timeout=240):
    env = dict(os.environ, AOMHIP_BENCH_RANKS_TIMEOUT_S="120", **(extra_env or {}))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "launcher_dry_run", "--dist-backend", "gloo",
                        "--steps", "2", "--warmup", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    return p, time.time() - t0


@pytest.mark.parametrize("n", [4, 8])
def test_child_process_launch_and_line_schema(n):
    p, _ = run(n)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1
    s = d["strong_scaling_search"]
    assert s["rccl_ranks_in_communicator"] == n and s["max_over_ranks_check"] == float(n)      # MAX over the ranks really saw every rank
    assert len(s["tile_columns_px"]) == n and sum(s["tile_columns_px"]) == 3840
    assert s["blocks_per_step"] == sum((w // 16) * (2160 // 16) for w in s["tile_columns_px"])   # SUM over the ranks
    ex = s["exchange"]["expected_bytes_per_rank_per_frame"]
    for mode in ("halo", "allgather"):
        assert len(ex[mode]["send"]) == n and len(ex[mode]["recv"]) == n
        assert sum(ex[mode]["send"]) == sum(ex[mode]["recv"]) > 0                                 # every byte sent is received by somebody
    # whole-column all-gather: a rank receives every column but its own
    assert ex["allgather"]["recv"] == [(3840 - w) * 2160 * 2 for w in s["tile_columns_px"]]
    assert max(ex["halo"]["recv"]) <= 2 * s["exchange"]["halo_px"] * 2160 * 2


def test_a_dead_rank_ends_the_job_with_a_non_zero_exit():
    p, el = run(4, {"AOMHIP_BENCH_FAIL_RANK": "2"})
    assert p.returncode != 0 and el < 100, (p.returncode, el)
    assert "terminating the others" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_a_short_communicator_ends_the_job_with_a_non_zero_exit():
    p, el = run(4, {"AOMHIP_BENCH_FAKE_COMM_RANKS": "3"})
    assert p.returncode != 0 and el < 100, (p.returncode, el)
    assert "communicator holds 3 ranks, the job has 4" in p.stderr


def test_the_driver_launch_form_under_torch_distributed_run():
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...`: bench.py then finds RANK / WORLD_SIZE in its environment and must NOT start ranks of its own."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29641",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "launcher_dry_run", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["strong_scaling_search"]["rccl_ranks_in_communicator"] == 2 and sum(d["strong_scaling_search"]["tile_columns_px"]) == 3840
