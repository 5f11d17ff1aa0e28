# SQ counter pass for one prof_render.py scene: bash tools/pmc_scene.sh <scene> <kernel substring> [frames]
scene=$1; kern=$2; frames=${3:-4}
root=$PWD; out=$root/gpurun_out/pmc_$scene
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- python3 $root/tools/prof_render.py megakernel $scene $frames > $out/run.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH --output-format csv -d $out/sq2 -- python3 $root/tools/prof_render.py megakernel $scene $frames >> $out/run.log 2>&1 || echo "second pass failed"
python3 $root/tools/pmc_summary.py $out $kern
