#!/usr/bin/env python3
"""markdown table of a bench.py line (main + legs) for DESIGN.md section 4: usage: bench_table.py bench.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def row(name, v):
    r = v["roofline"]
    cb = v.get("cpu_baseline") or {}
    f = lambda x, n=3: "—" if x is None else f"{x:.{n}f}"
    cpu = "—" if not cb.get("value") else "%.3g" % cb["value"]
    print(f"| {name} | {v['ms_per_step']:.3f} | {v['value']:.3g} | {f(r.get('kernel_ms'))} | {f(r.get('algorithmic_frac'))} | "
          f"{r.get('bound')} {f(r.get('frac'))} | {f(r.get('valu_counted_frac'))} / {f(r.get('lds_counted_frac'))} | "
          f"{f((r.get('hbm') or {}).get('traffic_frac'))} | {cpu} ({cb.get('cores', '—')}) |")


print("| workload | step ms | halos/s | dominant kernel ms | algorithmic frac (8 TB/s) | bound, frac | counted VALU / LDS | measured HBM traffic / peak | cpu_baseline halos/s (threads) |")
print("|---|---|---|---|---|---|---|---|---|")
row("headline", d)
for k, v in d.get("legs", {}).items():
    if isinstance(v, dict) and "roofline" in v:
        row(k, v)
        if "deposit_roofline" in v:
            r = v["deposit_roofline"]
            print(f"|   ↳ deposit | — | — | {r['kernel_ms']:.3f} | {r['algorithmic_frac']:.3f} | {r['bound']} {r['frac'] if r['frac'] is None else round(r['frac'], 3)} | "
                  f"{r.get('valu_counted_frac')} / {r.get('lds_counted_frac')} | — | — |")
for k in ("published", "multi_model"):
    v = d.get("legs", {}).get(k) or {}
    for q in ("paint", "baryonify", "multi_model"):
        if q in v:
            print(f"\n{k}.{q}: " + json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in v[q].items() if not isinstance(b, dict)}))
