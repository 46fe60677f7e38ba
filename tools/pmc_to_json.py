#!/usr/bin/env python3
"""profiles/pmc_traffic.json from rocprofv3 --pmc summaries (tools/pmc_summary.py output): HBM-side bytes per launch of the
dominant kernel of each measured workload = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B.  FETCH_SIZE is doubled per the gfx950
wide-read correction of MI355X_MICROARCH.md (an upper bound: not every read of these kernels is a wide coalesced stream);
WRITE_SIZE is taken as is.
usage: pmc_to_json.py TAG out.json  key=FETCH.txt,WRITE.txt[,kernel] ...   (key = bench.py's <workload>_<variant>_n<halos>_nside<nside>)"""
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


tag, out = sys.argv[1], sys.argv[2]
j = {"_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 3` on the {tag} build; summaries "
                f"profiles/{tag}_pmc_<workload>_<counter>.txt",
     "_note": "dominant kernel (shell_tile_kernel) per launch: (2*FETCH_SIZE + WRITE_SIZE)*1024 B; FETCH_SIZE doubled per the gfx950 "
              "wide-read correction of MI355X_MICROARCH.md (an upper bound)", "_raw": {}}
for spec in sys.argv[3:]:
    key, files = spec.split("=")
    parts = files.split(",")
    kernel = parts[2] if len(parts) > 2 else "shell_tile_kernel"
    fetch = write = 0.0
    nf = nw = 0
    for kern in kernel.split("+"):                             # "a+b+c": the sum over several kernels of one step (the tiled deposit)
        f_, nf = mean_of(parts[0], kern, "FETCH_SIZE")
        w_, nw = mean_of(parts[1], kern, "WRITE_SIZE")
        fetch, write = fetch + f_, write + w_
    if key.startswith("_"):                                    # not a bench key: kept under _raw only (e.g. the prep kernel)
        j["_raw"][key] = {"kernel": kernel, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "launches_averaged": [nf, nw]}
        continue
    j[key] = (2.0 * fetch + write) * 1024.0
    j["_raw"][key] = {"kernel": kernel, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "launches_averaged": [nf, nw]}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j))
