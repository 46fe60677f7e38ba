#!/bin/bash
# registers / scratch / occupancy / LDS of every kernel (build container; no GPU needed):  tools/kres.sh [EXTRA hipcc flags]
# output: one line per kernel, filtered by $KRES_FILTER (default: the shell / prep / regrid kernels)
src=/root/repo/baryonforge_amd/csrc/bfg_mi355.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -ffp-contract=on "$@" \
      -Rpass-analysis=kernel-resource-usage -o /tmp/kres_probe.so "$src" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {ag=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} / SGPRs:/ {sg=$(NF-1)}
       /LDS Size/ {printf "VGPR %-4s AGPR %-4s scratch B/lane %-5s occ %-3s LDS %-6s  %s\n", v, ag, s, o, $(NF-1), name}' |
  c++filt | grep -E "${KRES_FILTER:-shell_tile|halo_prep|regrid|small}" | cut -c1-200
