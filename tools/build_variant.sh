# Builds a variant of the library into build_variants/<name>/: bash tools/build_variant.sh <name> "<extra hipcc flags>"
name=$1; shift
mkdir -p build_variants/$name
make -C ray_tracer_amd/csrc -j8 OBJDIR=../../build_variants/$name OUT=../../build_variants/$name/librt_hip.so TOOL=../../build_variants/$name/rt_headless EXTRA="$*" 2>&1 | grep -E "error|warning: v|Error" ; ls -la build_variants/$name/librt_hip.so | cut -c20-
