# rocprofv3 evidence for every bench.py workload: bash tools/profile_configs.sh <tag> [configs]
# (run on the GPU box from the repository root; writes gpurun_out/prof_<tag>/)
#   bench_cfgN.json         the JSON line of `python3 bench.py --config N`, run on its own
#   kernel_stats_cfgN.csv   rocprofv3 --kernel-trace --stats of the same command
#   bench_cfgN.traced.json  the JSON line of that traced run (the tracer costs launches of many
#                           small kernels a few per cent)
#   pmc_cfgN.txt            counter passes of the dominant kernel (one rocprofv3 run per pass)
tag=${1:-x}
configs=${2:-"1 2 3 4"}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in $configs; do
  timeout -k 10 400 python3 $root/bench.py --config $c > $out/bench_cfg$c.json 2> $out/bench_cfg$c.err || exit 1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 $root/bench.py --config $c --no-cpu-baseline > $out/bench_cfg$c.traced.json 2>> $out/bench_cfg$c.err || exit 1
  find $out/stats_$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_cfg$c.csv
  rm -rf $out/stats_$c
done
export RT_TILE_PROBE=0   # counter passes: no probe launch among the dispatches that are averaged
for c in $configs; do
  kern=k_mega_regen; [ $c = 2 ] && kern=k_mega_sched; [ $c = 3 ] && kern=k_mega_sched
  pass() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$c/$name -- python3 $root/bench.py --config $c --steps 4 --warmup 1 --no-side-runs --no-cpu-baseline > /dev/null 2>> $out/pmc.err || echo "pass $name of config $c failed" >> $out/failed.txt
  }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
  pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
  pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  pass l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
  python3 $root/tools/pmc_summary.py $out/pmc_$c $kern > $out/pmc_cfg$c.txt
  rm -rf $out/pmc_$c
done
for c in $configs; do cut -c1-150 $out/bench_cfg$c.json; done
[ -f $out/failed.txt ] && cat $out/failed.txt
exit 0
