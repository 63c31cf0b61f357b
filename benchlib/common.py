"""Helpers every workload module shares: the clock ramp, the per-launch HIP-event timing and the roofline constants."""
import json
import os
import time

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


BD_OVERRIDE = 0   # bench.py --bit-depth: the search / filter workloads on planes of this depth (0 = their own)


def ramp(ctx, fn, seconds=None):
    """Untimed: keep the chip busy with the workload itself before anything is measured.  The first milliseconds after an idle period run
    at a lower clock (profiles/r03_sad_strip.md section 4: the same launch 0.310 ms right after 3 warm-up launches, 0.273 ms sustained);
    the W warm-up steps of the contract (a few hundred microseconds here) do not cover that."""
    seconds = float(os.environ.get("AOMHIP_BENCH_RAMP_S", "0.25")) if seconds is None else seconds
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        ctx.sync()


def kernel_avg_ms(ctx, fn, reps):
    """Average duration of `fn`'s launches: HIP events on the context's own stream (the stream the kernels are launched on)."""
    ramp(ctx, fn, 0.1)
    fn()
    ctx.sync()
    ctx.timer_begin()
    for _ in range(reps):
        fn()
    return ctx.timer_end() / reps


def load_traffic_entry(root, name, sha):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json); None when not measured or measured on another version
    of the kernel source (`sha` = sha256[:16] of the sources the figure describes)."""
    import json
    try:
        t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
        if (t.get("_measured_on") or {}).get(name) != sha:
            return None
        return t.get(name)
    except Exception:
        return None


def source_sha(root, files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(root, "aom-av1-psy_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# run-wide settings (bench.py's arguments), read by the workload modules
FRAMES_OVERRIDE = 0
TILE_COLUMNS = "uniform"   # --tile-columns
BACKEND = "nccl"           # RCCL; "gloo" only for dry runs of the N > 1 code path


TRAFFIC_SOURCES = {"sb": ("sad_sb.hip",), "sad": ("sad.hip",), "txq": ("xform_quant.hip", "txfm_device.h", "quant_device.h")}


def traffic_kind(name):
    return "sb" if name.endswith(":sb") else "txq" if name.startswith("txq") else "sad"


def kernel_source_sha(kind):
    """sha256[:16] of the kernel source a traffic figure describes (tools/pmc_traffic*.py store it beside the figure)."""
    import hashlib
    h = hashlib.sha256()
    for f in TRAFFIC_SOURCES[kind]:
        with open(os.path.join(ROOT, "aom-av1-psy_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_traffic(name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json,
    produced by tools/pmc_traffic*.py from separate rocprofv3 --pmc runs); None when not measured OR when the
    figure was measured on another version of the kernel source than the one in this tree (a stale counter is not evidence)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(p))
        if (t.get("_measured_on") or {}).get(name) != kernel_source_sha(traffic_kind(name)):
            return None
        return t.get(name)
    except Exception:
        return None
