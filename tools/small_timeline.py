#!/usr/bin/env python3
"""Timeline of ONE solve of a launch-bound workload out of a rocprofv3 --kernel-trace --memory-copy-trace CSV pair:
start offset, duration and the gap to the previous operation, for the last complete step of the run.
usage: small_timeline.py <kernel_trace.csv> [<memory_copy_trace.csv>] [first-kernel-substring]"""
import csv
import sys

ops = []
for r in csv.DictReader(open(sys.argv[1])):
    ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
if len(sys.argv) > 2 and sys.argv[2].endswith(".csv"):
    for r in csv.DictReader(open(sys.argv[2])):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
first = sys.argv[3] if len(sys.argv) > 3 else "k_chunk_init"
ops.sort()
starts = [i for i, o in enumerate(ops) if first in o[2]]
a, b = starts[-2], starts[-1]
t0 = ops[a][0]
prev_end = t0
for s, e, name in ops[a - 3:b]:
    print("%9.2f us  dur %7.2f  gap %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = e
