#!/usr/bin/env python3
"""Timeline of the configs[4] frame from a rocprofv3 --kernel-trace of `bench.py --workload inner_loop_4k_10bit`: the last N frames of the
timed region, per kernel the in-chain duration and the gap to the previous kernel's end -> markdown on stdout.
    python3 tools/gpu_timeline.py <dir with *_kernel_trace.csv> [frames=20]"""
import csv, glob, re, sys
from collections import defaultdict

d = sys.argv[1]
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda k: re.sub(r"\(.*", "", k.replace("void ", "").replace("aomhip::", "").replace("(anonymous namespace)::", ""))[:60]
# a frame = the kernels from one fullpel_diamond launch to the next; take the frames whose kernel sequence is the most common one
idx = [i for i, r in enumerate(rows) if "fullpel_diamond_kernel" in r[2]]
frames = [rows[a:b] for a, b in zip(idx[:-1], idx[1:])]
seqs = defaultdict(list)
for fr in frames:
    seqs[tuple(short(r[2]) for r in fr)].append(fr)
seq, frs = max(((k, v) for k, v in seqs.items() if len(k) >= 4), key=lambda kv: len(kv[1]))
frs = frs[-nfr:]
print("frames with the sequence of %d kernels: %d (of %d); using the last %d\n" % (len(seq), len(seqs[seq]), len(frames), len(frs)))
dur = defaultdict(list); gap = defaultdict(list)
for fi, fr in enumerate(frs):
    for i, (s, e, k) in enumerate(fr):
        dur[i].append((e - s) / 1e3)
        if i > 0:
            gap[i].append((s - fr[i - 1][1]) / 1e3)
    # gap of the first kernel = to the last kernel of the frame before (when that frame is the one right before in the trace)
period = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(frs[:-1], frs[1:]) if b[0][0] - a[0][0] < 5e6]
first_gap = [(b[0][0] - a[-1][1]) / 1e3 for a, b in zip(frs[:-1], frs[1:]) if b[0][0] - a[0][0] < 5e6]
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
print("| # | kernel | in-chain duration us (median) | gap before it us (median) |\n|---|---|---|---|")
tot_d = tot_g = 0.0
for i, k in enumerate(seq):
    g = med(first_gap) if i == 0 else med(gap[i])
    tot_d += med(dur[i]); tot_g += g
    print("| %d | `%s` | %.1f | %.1f |" % (i, k, med(dur[i]), g))
print("| | sum | %.1f | %.1f |" % (tot_d, tot_g))
print("\nframe period (start to start), median: %.1f us = %.0f frames/s; kernels %.1f us + gaps %.1f us" % (med(period), 1e6 / med(period), tot_d, tot_g))
