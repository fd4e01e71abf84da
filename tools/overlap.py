#!/usr/bin/env python3
"""Concurrency of the post-processing kernels in a rocprofv3 kernel trace of
tools/bench_pipeline.py: between two score-scan launches (given by their ordinal numbers), how
long each kernel type ran, how long at least one kernel ran, and how many ran at once."""
import csv
import glob
import sys

t = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(t))]
rows.sort()
lo, hi = int(sys.argv[2]), int(sys.argv[3])          # window: from the lo-th to the hi-th score scan launch
scans = [r for r in rows if 'score_scan' in r[2]]
a, b = scans[lo][0], scans[hi][0]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
per = {}
for s, e, n in sel:
    k = n.split('(')[0][-28:]
    per.setdefault(k, [0, 0])
    per[k][0] += e - s
    per[k][1] += 1
events = sorted([(s, 1) for s, e, n in sel] + [(e, -1) for s, e, n in sel])
busy = depth = 0
last = None
hist = {}
for ts, d in events:
    if last is not None and depth > 0:
        busy += ts - last
        hist[depth] = hist.get(depth, 0) + ts - last
    depth += d
    last = ts
span = sel[-1][1] - sel[0][0]
print(f"window {span / 1e3:.0f} us, kernels {len(sel)}, busy {busy / 1e3:.0f} us ({busy / span:.2f}), "
      f"sum of durations {sum(e - s for s, e, n in sel) / 1e3:.0f} us")
for k, (d, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:28s} calls {c:5d} avg {d / c / 1e3:7.2f} us  share of window {d / span:.2f}")
print("  time with N kernels running:", {k: round(v / span, 2) for k, v in sorted(hist.items())})
