set -e
for v in base g0 cur; do
  if [ $v = cur ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so; fi
  timeout -k 10 120 python tools/ab_lib.py 1920 1080
  timeout -k 10 120 python tools/ab_lib.py 3840 2160
done
