#!/usr/bin/env python3
"""profiles/sq_counters.json from rocprofv3 --pmc summaries (tools/pmc_summary.py output, one file per workload with all passes
appended): per bench key the dominant kernel's SQ counters per launch and two derived occupancies,
  valu_issue_frac = SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / kernel cycles      (quad-cycles -> cycles; one issue slot per VALU instruction)
  lds_pipe_frac   = SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles               (cycles in which a CU's LDS pipe is busy, conflicts included)
with kernel cycles = SQ_BUSY_CYCLES / 32 shader engines (the normalisation of profiles/r03_sq_counters.txt).
usage: sq_to_json.py TAG out.json key=summary.txt,kernel-substring ..."""
import json
import re
import sys


def counters_of(path, kernel):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        if cur is not None and kernel in cur:
            m = re.match(r"\s+(\w+)\s+n=\s*(\d+)\s+mean=([0-9.e+\-]+)", line)
            if m:
                out[m.group(1)] = float(m.group(3))
    return out


tag, out = sys.argv[1], sys.argv[2]
j = {"_source": f"rocprofv3 --pmc SQ_* (separate passes, --kernel-trace only) of `bench.py --steps 3 --warmup 1` on the {tag} build; "
                f"summaries profiles/{tag}_sq_counters_<workload>.txt (tools/r06_profile.sh sq)",
     "_note": "per launch of the named kernel; valu_issue_frac = SQ_ACTIVE_INST_VALU*4/1024/(SQ_BUSY_CYCLES/32), "
              "lds_pipe_frac = SQ_LDS_IDX_ACTIVE/256/(SQ_BUSY_CYCLES/32)"}
for spec in sys.argv[3:]:
    key, rest = spec.split("=")
    path, kernel = rest.split(",", 1)
    c = counters_of(path, kernel)
    if "SQ_BUSY_CYCLES" not in c:
        print(f"{key}: no counters of {kernel} in {path}", file=sys.stderr)
        continue
    cyc = c["SQ_BUSY_CYCLES"] / 32.0
    e = {"kernel": kernel, "kernel_cycles": cyc}
    if "SQ_ACTIVE_INST_VALU" in c:
        e["valu_issue_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
    if "SQ_LDS_IDX_ACTIVE" in c:
        e["lds_pipe_frac"] = c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        e["wave_wait_share"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    e["raw"] = c
    j[key] = e
json.dump(j, open(out, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "raw"} for k, v in j.items() if not k.startswith("_")}, indent=1))
