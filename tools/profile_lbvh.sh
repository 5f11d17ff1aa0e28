# Kernel-time and counter evidence for the LBVH scenes: bash tools/profile_lbvh.sh <tag>
# (run on the GPU box from the repository root; writes gpurun_out/lbvh_<tag>/)
#   stats_grid     rocprofv3 --kernel-trace --stats of the 10k-sphere scene (build + k_mega_sched)
#   stats_terrain  the same for the 1,002,528-triangle terrain, rebuilt every frame
#   pmc_<scene>    counter passes of k_mega_sched, one rocprofv3 run per pass
tag=${1:-x}
root=$PWD
out=$root/gpurun_out/lbvh_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export RT_TILE_PROBE=0
run_stats() {  # name, then the program
  name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$name -- "$@" > $out/stats_$name.log 2>&1 || return 1
  find $out/stats_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_$name.csv
  rm -rf $out/stats_$name
}
run_pmc() {  # scene, pass name, counters...
  scene=$1; pass=$2; shift 2
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$scene/$pass -- python3 $root/tools/prof_render.py megakernel $scene 4 >> $out/pmc_$scene.log 2>&1 || echo "pass $pass of $scene failed" >> $out/failed.txt
}
run_stats grid python3 $root/tools/prof_render.py megakernel grid 6 || exit 1
run_stats terrain python3 $root/tools/cfg4_terrain.py || exit 1
for scene in grid terrain; do
  run_pmc $scene sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
  run_pmc $scene sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
  run_pmc $scene l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  run_pmc $scene l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
  run_pmc $scene fetch FETCH_SIZE
  run_pmc $scene write WRITE_SIZE
  python3 $root/tools/pmc_summary.py $out/pmc_$scene k_mega_sched > $out/pmc_$scene.txt
  rm -rf $out/pmc_$scene
done
cat $out/pmc_grid.txt $out/pmc_terrain.txt
[ -f $out/failed.txt ] && cat $out/failed.txt
exit 0
