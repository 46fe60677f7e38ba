#!/bin/bash
# GPU box: the default bench.py run (main line + every leg) into gpurun_out/r06/<tag>.json, with a short summary
tag=${1:-bench}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
t0=$(date +%s)
python3 bench.py > gpurun_out/r06/$tag.json 2> gpurun_out/r06/$tag.err
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -c 1500 gpurun_out/r06/$tag.err
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r06/$tag.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("main", d["value"], round(d["ms_per_step"], 4), r["bound"], r["frac"], "alg", r["algorithmic_frac"], "valu", r["valu_counted_frac"], "lds", r["lds_counted_frac"],
      "cpu", (d.get("cpu_baseline") or {}).get("value"))
for k, v in d["legs"].items():
    if not isinstance(v, dict) or "error" in v:
        print(k, v); continue
    r = v["roofline"]
    print(k, round(v["ms_per_step"], 4), r["bound"], r["frac"], "alg", r.get("algorithmic_frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"),
          (v.get("cpu_baseline") or {}).get("error"), "wall", round(v["leg_wall_s"], 1))
    if "deposit_roofline" in v: print("   deposit", v["deposit_roofline"]["bound"], v["deposit_roofline"]["frac"], v["deposit_roofline"]["algorithmic_frac"])
    for q in ("paint", "baryonify"):
        if q in v: print("  ", q, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v[q].items() if not isinstance(b, (dict, str))})
PY
