# GPU box: low-perturbation item timeline (three -DBFG_STAGE_TIMING=4 builds under build/, timing threads 0 / 64 / 448)
for n in 100000 10000 1000000; do
  for t in 0 64 448; do
    BFG_ST_MODE=4 BFG_ST_TID=$t BFG_SO=$PWD/build/bfg_st4_$t.so python3 tools/stage_timing.py $n 1024 paint 2>&1 | grep -v "^/opt\|warn"
  done
done
