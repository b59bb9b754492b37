#!/usr/bin/env python3
"""Kernel timeline of a few consecutive pipelined steps from a `rocprofv3 --kernel-trace` CSV
(tools/profile.sh -> gpurun_out/prof_<tag>/trace/*/*_kernel_trace.csv): start, end and duration of every
kernel relative to the first site pass shown, the queue it ran on, and the idle gap on the main queue
between two site passes.  Usage: python tools/timeline.py <kernel_trace.csv> [first_step] [steps]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+)_kernel\b", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"].split("(")[0][-40:]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
passes = [i for i, r in enumerate(rows) if r[2].startswith("site_counts")]
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(len(passes) - 6, 0)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sel = passes[first : first + n + 1]
if len(sel) < 2:
    sys.exit("not enough site passes in the trace")
t0 = rows[sel[0]][0]
queues = {}
print(" start_us    end_us  dur_us  kernel (queue)")
for s, e, name, q in rows[sel[0] : sel[-1]]:
    qi = queues.setdefault(q, len(queues))
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {name} ({qi})")
starts = [rows[i][0] for i in sel]
ends = [rows[i][1] for i in sel]
print()
print("site pass to site pass: " + ", ".join(f"{(b - a) / 1e3:.1f}" for a, b in zip(starts, starts[1:])) + " us; "
      "idle gap on the main queue between two site passes: " + ", ".join(f"{(b - a) / 1e3:.1f}" for a, b in zip(ends, starts[1:])) + " us")
