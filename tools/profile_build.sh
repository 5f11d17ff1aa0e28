# Kernel trace of the terrain workload with one frame in flight (rebuild, then trace, nothing beside them):
#   bash tools/profile_build.sh <tag>      (on the GPU box, from the repository root)
tag=${1:-x}
root=$PWD
out=$root/gpurun_out/prof_build_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --config 3 --images-in-flight 1 --steps 40 --warmup 5 --no-side-runs --no-cpu-baseline > $out/bench.json 2> $out/bench.err || exit 1
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats
cut -c1-200 $out/bench.json
