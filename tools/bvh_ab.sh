# 10k-sphere and terrain frame times of library variants: bash tools/bvh_ab.sh <variant|cur>...
# (BVH_SIZE="W H" overrides 1920 1080)
for v in "$@"; do
  if [ $v = cur ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so; fi
  echo "$v grid: $(LONE_TILES=0,0 LONE_SCENE=grid timeout -k 10 120 python tools/lone_tile.py $BVH_SIZE | tr '\n' ' ')"
  echo "$v terrain: $(LONE_TILES=0,0 LONE_SCENE=terrain timeout -k 10 120 python tools/lone_tile.py $BVH_SIZE | tr '\n' ' ')"
done
