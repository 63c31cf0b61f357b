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
    eb = ex["expected_bytes_per_frame_max_rank"]   # from the exchange plan alone: what the driver's SCALE record can be checked against
    assert set(eb) == {"halo_send", "halo_recv", "allgather_send", "allgather_recv"} and all(v == 0 for v in eb.values())   # one rank: nothing to exchange
    assert s["tile_columns_px"] == [3840] and s["tile_columns_px_uniform"] == [3840] and s["tile_columns_px_balanced"] == [3840]
    assert s["single_gpu_same_box"]["value"] > 0 and s["speedup_over_single_gpu"] > 0
    assert s["parity_sample_slot0"] in (True, None)


def test_default_single_gpu_line_is_small_enough_for_the_driver_record(tmp_path):
    """The driver keeps an 8 KB tail of the output: round 4's 31 KB default line was not parsed.  The default N = 1 run must end in ONE JSON
    line under 8000 bytes that still carries the headline, the roofline (flat scalars) and the cpu baseline; everything else is in the side file."""
    full = tmp_path / "bench_full.json"
    env = dict(os.environ, AOMHIP_BENCH_RAMP_S="0.02", AOMHIP_BENCH_FULL=str(full), AOMHIP_BENCH_CPU_SECONDS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    out_lines = p.stdout.splitlines()
    assert len([ln for ln in out_lines if ln.startswith("{")]) == 1, "exactly ONE JSON line on stdout"
    last = out_lines[-1]
    assert len(last) < 8000, len(last)
    d = json.loads(last)
    assert d["metric"] == "SAD-candidates/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["workload"] == "sad16x16_modeA_1080p_8bit"
    r = d["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in r.values()), "roofline: flat scalars only"
    assert 0 < r["frac"] < 1 and r["peak"] == 8000.0 and r["bound"] in ("hbm", "mfma") and r["achieved"] > 0 and "traffic" in r
    assert 0 < r["frac_4k_8bit"] < 1 and 0 < r["frac_4k_10bit"] < 1 and r["avg_launch_ms_4k_8bit"] > 0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and c["sample"]
    assert set(d["txq"]) == {"fwd_txfm2d+quantize_b_1080p_8bit", "fwd_txfm2d+quantize_b_4k_10bit"}
    for t in d["txq"].values():
        assert t["value"] > 0 and 0 < t["roofline"]["frac"] < 1 and set(t["per_size_frac"]) == {"4x4", "8x8", "16x16", "32x32"}
    assert d["parity_all"] is True and d["parity_frame0_and_last_slot"] is True
    assert "encode_inner_loop_4k_10bit" in d["others"]
    # round 6: the measurement gaps of SURVEY 8(d) closed IN THE LINE, which still stays under 6 KB
    assert len(last) < 6200, len(last)
    v = d["variance"]
    assert set(v) == {"var_1080p_8bit", "subpel_var_1080p_8bit", "var_4k_10bit", "subpel_var_4k_10bit"}
    for e in v.values():   # frac: algorithmic bytes (520 / 553 / 1 032 / 1 098 B per evaluation), c: compulsory bytes, t: counter traffic
        assert e["frac"] > 0 and 0 < e["c"] < 1 and e["ms"] > 0 and e["parity"] is True and "t" in e
    for k in ("var_1080p_8bit", "var_4k_10bit"):   # the full-pel lists also through the strip walk (aomhip_variance_sb_batch), bit-identical and faster
        assert v[k]["sb_same"] is True and 0 < v[k]["sb_ms"] < v[k]["ms"] and v[k]["c"] < v[k]["sb_c"] < 1 and "sb_t" in v[k]
    fr = d["filters_ring"]
    for k in ("deblock_vert+horz", "cdef_luma"):   # on a 1.3 GB ring of 4K 10-bit planes: an HBM figure
        assert 0 < fr[k]["frac"] < 1 and 0 < fr[k]["c"] < 1 and fr[k]["us"] > 0 and "t" in fr[k]
    assert fr["parity"] is True and fr["ring_GB"] >= 1.0
    # one trial of the loop-filter level search (copy + deblock + plane SSE): tens of microseconds -- round 1's plane SSE alone was 392 us of serialised atomics
    assert 0 < fr["lpf_trial_us"] < 150
    il = d["others"]["encode_inner_loop_4k_10bit"]
    assert 0 < il["roofline_frac"] < 1 and 0 < il["yuv420_roofline_frac"] < 1 and il["yuv420_ms_per_frame"] > il["ms_per_frame"]
    for t in d["txq"].values():
        assert set(t["qindex_frac_16x16"]) == {"20", "100", "200"} and all(0 < x < 1 for x in t["qindex_frac_16x16"].values())
        assert 0 < t["tx_type_frac_16x16"]["min"] <= t["tx_type_frac_16x16"]["max"] < 1 and "traffic" in t["roofline"]
    # the full record: every informational workload entire, the per-size roofline table, the cpu legs
    f = json.loads(full.read_text())
    assert len(f["others"]) >= 20 and "sizes" in f["roofline"] and "legs" in f["cpu_baseline"]
    # ... and each of them as one JSON line on stderr
    assert sum(1 for ln in p.stderr.splitlines() if ln.startswith("{")) == len(f["others"])
