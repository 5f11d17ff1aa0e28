# Kernel times and counters of the wavefront pipeline with and without ray regrouping:
#   bash tools/profile_regroup.sh <scene: terrain|grid>     (on the GPU box, from the repository root)
scene=${1:-terrain}
root=$PWD
out=$root/gpurun_out/prof_regroup_$scene
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for rg in 0 1; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$rg -- python3 $root/tools/wf_regroup.py $scene $rg > $out/run_$rg.txt 2>> $out/err.txt || exit 1
  find $out/stats_$rg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_regroup$rg.csv
  rm -rf $out/stats_$rg
  timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $out/pmc_$rg/sq -- python3 $root/tools/wf_regroup.py $scene $rg > /dev/null 2>> $out/err.txt || exit 1
  timeout -s KILL 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_$rg/mem -- python3 $root/tools/wf_regroup.py $scene $rg > /dev/null 2>> $out/err.txt || exit 1
  python3 $root/tools/pmc_summary.py $out/pmc_$rg k_wf_ > $out/pmc_regroup$rg.txt
  rm -rf $out/pmc_$rg
done
cat $out/run_0.txt $out/run_1.txt
