"""bench.py's N > 1 line, end to end on ONE GPU: RCCL refuses two ranks on one device, so the multi-GPU code path runs with a process group
and an RCCL communicator of one rank (AOMHIP_BENCH_FORCE_DIST=1) -- torch.distributed set-up, the unique-id broadcast, aomhip_comm_init,
aomhip_allgather_recon in front of every search step, the reductions, the single-GPU comparison leg.  What the driver's SCALE run must find in
the line is asserted here: the same headline metric at every N, the strong-scaling search block with both exchange modes timed, the number
of ranks RCCL really holds, and the tile-column widths under both of the reference's rules."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tile_columns", ["uniform", "balanced"])
def test_forced_one_rank_run_emits_the_multi_gpu_schema(tile_columns):
    env = dict(os.environ, AOMHIP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", AOMHIP_BENCH_RAMP_S="0.02")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--frames-per-gpu", "8",
                        "--tile-columns", tile_columns], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line"
    d = json.loads(lines[0])
    assert d["metric"] == "SAD-candidates/s" and d["unit"] == "candidates/s" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["data"] == "synthetic" and d["dtype"] == "u8"
    assert d["config"]["workload"] == "sad16x16_modeA_1080p_8bit" and d["vs_baseline"] is None
    assert tile_columns in d["config"]["partition"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "ceiling_GBs", "frac_of_ceiling"):
        assert k in r, k
    assert 0 < r["frac"] < 1 and r["peak"] == 8000.0
    s = d["strong_scaling_search"]
    assert s is not None and "error" not in s, s
    assert s["metric"] == "search blocks/s" and s["scaling"] == "strong" and s["value"] > 0
    assert s["rccl_ranks_in_communicator"] == d["n_gpus"]                     # every rank of the job really joined the communicator
    ex = s["exchange"]
    assert ex["halo_ms_per_frame"] > 0 and ex["allgather_ms_per_frame"] > 0   # both exchange modes were run and timed
    assert ex["mode"] == "halo" and ex["halo_px"] == 132
    assert s["tile_columns_px"] == [3840] and s["tile_columns_px_uniform"] == [3840] and s["tile_columns_px_balanced"] == [3840]
    assert s["single_gpu_same_box"]["value"] > 0 and s["speedup_over_single_gpu"] > 0
    assert s["parity_sample_slot0"] in (True, None)
