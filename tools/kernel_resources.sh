# VGPRs and scratch bytes of every kernel of one source file: bash tools/kernel_resources.sh rt_megakernel.hip [extra flags]
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "$@" -S --cuda-device-only -o /tmp/kr.s ray_tracer_amd/csrc/$src 2>/dev/null
grep -E "^\s+\.set .*\.(num_vgpr|private_seg_size), " /tmp/kr.s | sed -E 's/^\s+\.set //; s/EEvNS_6ParamsENS_5SceneENS_8FrameSeqENS_7OutputsE//; s/_ZN2rt[0-9]+//' | paste - - | awk '{print $1, $2, $4}' | sed 's/\.num_vgpr,//'
