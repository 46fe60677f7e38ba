#!/usr/bin/env python3
"""Per-kernel durations and the gaps between consecutive kernels of a step, from a rocprofv3 --kernel-trace CSV
(usage: step_gaps.py <..._kernel_trace.csv> [first kernel of a step, default halo_prep_kernel])"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "halo_prep_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bfg::", "")[:60]
steps, cur = [], []
for r in rows:
    if first in r["Kernel_Name"] and cur:
        steps.append(cur); cur = []
    cur.append(r)
steps.append(cur)
steps = [s for s in steps if first in s[0]["Kernel_Name"]]
steps = steps[len(steps) // 2:]                      # the second half: steady state
sig = collections.Counter(tuple(name(r) for r in s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if tuple(name(r) for r in s) == sig]
print(f"{len(steps)} steps of {len(sig)} kernels")
tot = 0.0
for k, nm in enumerate(sig):
    dur = sum(int(s[k]["End_Timestamp"]) - int(s[k]["Start_Timestamp"]) for s in steps) / len(steps) / 1e3
    if k + 1 < len(sig):
        gap = sum(int(s[k + 1]["Start_Timestamp"]) - int(s[k]["End_Timestamp"]) for s in steps) / len(steps) / 1e3
    else:
        nxt = [(int(b[0]["Start_Timestamp"]) - int(a[-1]["End_Timestamp"])) for a, b in zip(steps, steps[1:])
               if int(b[0]["Start_Timestamp"]) - int(a[-1]["End_Timestamp"]) < 1e6]
        gap = sum(nxt) / max(len(nxt), 1) / 1e3
    tot += dur + gap
    print(f"  {nm:60s} {dur:8.2f} us   gap after {gap:6.2f} us")
print(f"  sum {tot:.2f} us")
