#!/usr/bin/env python3
"""top kernels of a rocprofv3 --stats kernel_stats.csv: calls, average us, share (usage: stats_top.py file.csv [n])"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'].split('(')[0][:58]:58s} {int(r['Calls']):5d} calls  {float(r['AverageNs'])/1e3:10.1f} us  {float(r['Percentage']):5.1f} %")
