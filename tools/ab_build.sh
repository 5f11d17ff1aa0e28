# LBVH rebuild time of the terrain for several library variants: bash tools/ab_build.sh "<variants>"
for rep in 1 2; do
  for v in $1; do
    lib=$PWD/build_variants/$v/librt_hip.so; [ $v = current ] && lib=$PWD/ray_tracer_amd/librt_hip.so
    RT_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --config 3 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['lbvh_build']; print('$v', 'build ms', round(b['ms'],4), 'min', round(b['ms_min'],4), 'step', round(d['ms_per_step'],3))"
  done
done
