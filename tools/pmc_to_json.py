#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two rocprofv3 --pmc summaries (tools/pmc_summary.py output) of the headline run:
HBM-side bytes per launch of the dominant kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B.  FETCH_SIZE is doubled per the
gfx950 wide-read correction of MI355X_MICROARCH.md (an upper bound: not every read of this kernel is a wide coalesced
stream); WRITE_SIZE is exact for 16-B-per-lane stores.   usage: pmc_to_json.py FETCH.txt WRITE.txt TAG [out.json]"""
import json
import re
import sys


def mean_of(path, kernel, counter):
    txt = open(path).read().splitlines()
    for i, line in enumerate(txt):
        if kernel in line:
            for l2 in txt[i + 1:i + 6]:
                m = re.search(counter + r"\s+n=\s*(\d+)\s+mean=([0-9.e+]+)", l2)
                if m:
                    return float(m.group(2)), int(m.group(1))
    raise SystemExit(f"{counter} of {kernel} not found in {path}")


fetch, nf = mean_of(sys.argv[1], "shell_tile_kernel", "FETCH_SIZE")
write, nw = mean_of(sys.argv[2], "shell_tile_kernel", "WRITE_SIZE")
tag = sys.argv[3]
out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
j = {"paint_auto_n1000000_nside1024": (2.0 * fetch + write) * 1024.0,
     "_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) of `bench.py --steps 3` on the {tag} build: "
                f"profiles/{tag}_pmc_FETCH_SIZE.txt, profiles/{tag}_pmc_WRITE_SIZE.txt",
     "_note": "shell_tile_kernel per launch: (2*FETCH_SIZE + WRITE_SIZE)*1024 B; FETCH_SIZE doubled per the gfx950 wide-read "
              "correction of MI355X_MICROARCH.md (an upper bound); WRITE_SIZE = the 101 MB map written once + the deferred-pixel lists",
     "_raw": {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "launches_averaged": [nf, nw]}}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j))
