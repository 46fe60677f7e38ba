#!/usr/bin/env python3
"""Summary of a tools/ab_so.sh log with two builds: per workload the kernel times of every round and the ratio of the means; the
builds' maps are compared by the sum / non-zero count / P_tot the first round prints.  usage: ab_summary.py log before.so after.so"""
import collections
import re
import sys

log, a_so, b_so = sys.argv[1:4]
d = collections.defaultdict(lambda: collections.defaultdict(list))
so = None
for line in open(log):
    if line.startswith("=="):
        so = line.split()[1]
        continue
    m = re.match(r"(\S+)\s+-\s+step\s+([\d.]+) ms\s+kernel\s+([\d.]+)", line)
    if not m:
        continue
    d[m.group(1)][so].append(float(m.group(3)))
    m2 = re.search(r"sum (\S+) nonzero (\d+) P_tot (\d+)", line)
    if m2:
        d[m.group(1)][so + "_chk"].append(m2.groups())
for w, v in d.items():
    a, b = v[a_so], v[b_so]
    same = ""
    if v.get(a_so + "_chk") and v.get(b_so + "_chk"):
        ca, cb = v[a_so + "_chk"][0], v[b_so + "_chk"][0]
        same = "  same nonzero / P_tot: %s, sums %s vs %s" % (ca[1:] == cb[1:], ca[0], cb[0])
    print("%-9s before %s   after %s   ratio %.3f%s" % (w, " ".join("%.3f" % x for x in a), " ".join("%.3f" % x for x in b),
                                                      (sum(b) / len(b)) / (sum(a) / len(a)), same))
