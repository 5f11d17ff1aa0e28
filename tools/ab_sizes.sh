# frame times of library variants: bash tools/ab_sizes.sh "<W H>" <variant|cur>...
size=$1; shift
for v in "$@"; do
  if [ $v = cur ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so; fi
  printf "%s %s: " "$size" $v
  timeout -k 10 100 python tools/ab_lib.py $size || exit 1
done
