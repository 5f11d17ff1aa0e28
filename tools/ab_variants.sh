# A/B of library variants built by tools/build_variant.sh: bash tools/ab_variants.sh <variant|cur>...
# (two alternations; default scene at 1920x1080 and 3840x2160, the LBVH scenes, the textured 8K batch)
for round in 1 2; do
for v in "$@"; do
  if [ $v = cur ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so; fi
  timeout -k 10 100 python tools/ab_lib.py 1920 1080 | sed "s/^[^ ]* /$v 1080p /" || exit 1
  timeout -k 10 100 python tools/ab_lib.py 3840 2160 | sed "s/^[^ ]* /$v 4K /" || exit 1
  timeout -k 10 200 python tools/lbvh_time.py grid glass terrain glass4k | sed "s/^[^ ]* /$v /" || exit 1
done
done
