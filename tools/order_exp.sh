cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
for cfg in off 12 13 off 12 13; do
  if [ $cfg = off ]; then unset RGCN_ITEMS_BY_BLOCK; else export RGCN_ITEMS_BY_BLOCK=$cfg; fi
  python3 bench.py --no-cpu-baseline --no-secondary > $out/r04o_bench_${cfg}_$RANDOM.json 2>/dev/null
  echo "bench $cfg done"
done
for cfg in 12 13; do
  export RGCN_ITEMS_BY_BLOCK=$cfg
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=$out/pmc_r04o_$cfg/$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > /dev/null 2> $out/r04o_pmc.err
    echo "pmc $cfg $set done"
  done
  python3 tools/pmc_summary.py $out/pmc_r04o_$cfg $out/r04o_pmc_counters_$cfg.json "C2 bench, RGCN_ITEMS_BY_BLOCK=$cfg" > $out/r04o_pmc_summary_$cfg.txt
done
