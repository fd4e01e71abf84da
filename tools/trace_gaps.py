#!/usr/bin/env python3
"""Gap structure of a rocprofv3 kernel trace: for every (previous kernel, next kernel) pair that
follows one another on the device, the median idle time between them and the kernels' median
durations.  usage: tools/trace_gaps.py <dir holding *kernel_trace.csv> [min_count]"""
import collections
import csv
import glob
import statistics
import sys


def short(n):
    n = n.split("(")[0]
    return n.split("rh::")[-1][:40] if "rh::" in n else n[-40:]


def main():
    f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
    min_count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    dur = collections.defaultdict(list)
    gap = collections.defaultdict(list)
    prev = None
    for r in rows:
        s, e, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
        dur[n].append((e - s) / 1e3)
        if prev is not None:
            gap[(prev[1], n)].append((s - prev[0]) / 1e3)
        prev = (e, n)
    print(f"{len(rows)} kernels in {f}")
    print("durations (us): median  p10  p90  count")
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= min_count:
            q = statistics.quantiles(v, n=10) if len(v) > 1 else [v[0]] * 9
            print(f"  {n:42s} {statistics.median(v):8.2f} {q[0]:8.2f} {q[-1]:8.2f} {len(v):6d}")
    print("gaps between consecutive kernels (us; negative = overlap): median  p10  p90  count")
    for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1])):
        if len(v) >= min_count:
            q = statistics.quantiles(v, n=10) if len(v) > 1 else [v[0]] * 9
            print(f"  {k[0]:34s} -> {k[1]:34s} {statistics.median(v):8.2f} {q[0]:8.2f} {q[-1]:8.2f} {len(v):6d}")


if __name__ == "__main__":
    main()
