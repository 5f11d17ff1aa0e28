for size in "1920 1080" "3840 2160"; do
  for v in cur w6 cur w6; do
    if [ $v = cur ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so; fi
    printf "%s mixed/32 %s: " "$size" $v
    RT_REGEN_FORM=mixed RT_REGEN_CHAIN_DIV=32 timeout -k 10 100 python tools/ab_lib.py $size || exit 1
  done
done
