# GPU box: item-level stage timing (three -DBFG_STAGE_TIMING=3 builds under build/, timing threads 64 / 0 / 448)
for t in 64 0 448; do
  for n in 100000 10000 1000000; do
    echo "== tid $t n $n"
    BFG_ST_MODE=3 BFG_ST_TID=$t BFG_SO=$PWD/build/bfg_st3_$t.so python3 tools/stage_timing.py $n 1024 paint 2>&1 | grep -v "^/opt\|warn"
  done
done
