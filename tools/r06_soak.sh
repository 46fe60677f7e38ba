#!/bin/bash
# GPU box: round 6 soak.  (1) LightconeShell(pinned="inplace") on page-aligned buffers (tests/soak/soak_inplace.py: VERDICT r5 item 6),
# (2) the randomised soak against the oracle on the round's build -- plan reuse (BFG_SHELL_REUSE_PLAN) is active inside every case that
# paints twice over one catalog --, default paths and a few A/B paths, (3) snapshot / deposit / grid.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
O=gpurun_out/r06/soak_${SOAK_PART:-all}.txt
: > $O
run() { d=$(( $2 * ${SOAK_SCALE:-1} )); sd=$(( $3 + ${SOAK_SEED:-0} )); echo "== $1 ($d s, seed $sd)" >> $O; env $1 BFG_SOAK_TRACE=gpurun_out/r06/soak_trace_$sd.txt timeout -k 10 $(( d + 120 )) python3 tests/soak/soak.py $d $sd 2>&1 | grep -v "^ok\|amdgpu.ids" | tail -3 >> $O; }
if [ "${SOAK_PART:-all}" != 2 ]; then
echo "== in-place page-locked shells (soak_inplace.py ${INPLACE_CASES:-20000} cases)" >> $O
timeout -k 10 ${INPLACE_TIMEOUT:-900} python3 tests/soak/soak_inplace.py ${INPLACE_CASES:-20000} 6001 2>&1 | grep -v "amdgpu.ids" | tail -4 >> $O
fi
if [ "${SOAK_PART:-all}" = 3 ]; then      # after the soak learnt to paint a second model on the first one's plan: default + forced slow binning path
run "BFG_X=0" 150 6201
run "BFG_TILE_CAP=3 BFG_TILE_SCAN=1" 90 6202
run "BFG_TILE_CAP=2 BFG_PAIR_CAP=100" 45 6203
run "BFG_D2H_SLICES=1 BFG_EAGER_SOA=1" 45 6204
run "BFG_TILE_LIGHT=1" 45 6205
cat $O; if grep -q "Memory access fault\|Error\|error\|dumped core" $O; then exit 1; fi; exit 0
fi
if [ "${SOAK_PART:-all}" != 1 ]; then
run "BFG_X=0" 200 6101
run "BFG_SOAK_INPLACE=all" 90 6102
run "BFG_TILE_CAP=3 BFG_TILE_SCAN=1" 60 6103
run "BFG_PLAN_REUSE=0" 60 6104
run "BFG_TILE_LIGHT=1" 45 6105
run "BFG_ND_FROM_DIM=4" 45 6106
echo "== callable models" >> $O
timeout -k 10 200 python3 tests/soak/soak_callable.py 45 6107 2>&1 | grep -v "^ok\|amdgpu.ids" | tail -3 >> $O
echo "== aux (snapshot / deposit / grid)" >> $O
timeout -k 10 200 python3 tests/soak/soak_aux.py 60 6108 2>&1 | grep -v "^ok\|amdgpu.ids" | tail -3 >> $O
fi
for f in gpurun_out/r06/soak_trace_*.txt; do [ -f "$f" ] && ! tail -1 "$f" | grep -q ": ok$" && { echo "== unfinished: $f" >> $O; tail -2 "$f" >> $O; }; done
cat $O
if grep -q "Memory access fault\|Error\|error\|dumped core" $O; then exit 1; fi
