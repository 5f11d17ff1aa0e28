# LBVH workloads for several library variants, alternating: bash tools/ab_lbvh.sh "<variants>" "<scenes of tools/lbvh_time.py>"
for rep in 1 2; do
  for v in $1; do
    RT_HIP_LIB=$PWD/build_variants/$v/librt_hip.so timeout -k 10 200 python tools/lbvh_time.py $2 2>&1 | grep -v amdgpu.ids
  done
done
