#!/bin/bash
# registers / scratch of the tile-kernel instantiations: tools/scratch.sh [source.hip] [out.so]   (build container; no GPU needed)
src=${1:-/root/repo/baryonforge_amd/csrc/bfg_mi355.hip}; out=${2:-/tmp/scratch_probe.so}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -ffp-contract=on ${EXTRA} \
      -Rpass-analysis=kernel-resource-usage -o "$out" "$src" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /ScratchSize/ {s=$0; sub(/.*lane\]: /,"",s); sub(/ .*/,"",s); if (name ~ /shell_tile|halo_prep|tile_fill/) print s, name}'
