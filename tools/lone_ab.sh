# lone-tile and whole-frame times of library variants: bash tools/lone_ab.sh <variant>...
export LONE_TILES=${LONE_TILES:-1,1,0,0}
for v in "$@"; do
  RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so timeout -k 10 120 python tools/lone_tile.py | tr "\n" " " || exit 1
  echo $v
done
