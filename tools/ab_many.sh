# Frame times of several library variants on the default scene, alternating, three rounds:
#   bash tools/ab_many.sh "<variant> <variant> ..." [w h]
vs=$1; shift
for rep in 1 2 3; do
  for v in $vs; do
    RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so timeout -k 10 120 python tools/ab_lib.py "$@" 2>&1 | grep -v amdgpu.ids
  done
done
