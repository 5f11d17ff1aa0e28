# A/B of two library variants on the default scene, alternating: bash tools/ab_two.sh <variantA> <variantB> [w h]
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  for v in $a $b; do
    RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so timeout -k 10 120 python tools/ab_lib.py "$@" 2>&1 | grep -v amdgpu.ids
  done
done
