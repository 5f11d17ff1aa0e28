# where the waves of a scene's kernel spend their time: bash tools/pmc_wait.sh <scene> <kernel substring>
scene=$1; kern=$2
root=$PWD; out=$root/gpurun_out/pmcw_$scene
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $out/a -- python3 $root/tools/prof_render.py megakernel $scene 4 > $out/run.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_IFETCH SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $out/b -- python3 $root/tools/prof_render.py megakernel $scene 4 >> $out/run.log 2>&1 || echo "second pass failed"
python3 $root/tools/pmc_summary.py $out $kern
