# Profiles of the benchmark kernel for profiles/: bash tools/profile_round.sh <tag>
# (run on the GPU box from the repository root; writes gpurun_out/prof_<tag>/)
tag=${1:-x}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err || exit 1
export RT_TILE_PROBE=0   # counter passes: no probe launch among the dispatches that are averaged
for size in 1920x1080 3840x2160; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_$size/fetch -- python3 $root/tools/prof_render.py megakernel default 8 $size > /dev/null 2>> $out/pmc.err || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_$size/write -- python3 $root/tools/prof_render.py megakernel default 8 $size > /dev/null 2>> $out/pmc.err || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$size/sq -- python3 $root/tools/prof_render.py megakernel default 8 $size > /dev/null 2>> $out/pmc.err || exit 1
  python3 $root/tools/pmc_summary.py $out/pmc_$size k_mega_regen > $out/pmc_$size.txt || exit 1
done
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats $out/pmc_1920x1080 $out/pmc_3840x2160
cat $out/bench.json | cut -c1-160; cat $out/pmc_*.txt
