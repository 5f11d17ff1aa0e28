set -e
cd $GRAFT_REPO_ROOT
echo "current mixed:"; timeout -k 10 120 python tools/ab_lib.py
for n in 2 16 64 256; do
  for form in mixed batched; do
    echo "skip $n form $form:"; RT_HIP_LIB=$PWD/build_variants/skip$n/librt_hip.so RT_REGEN_FORM=$form timeout -k 10 120 python tools/ab_lib.py
  done
done
echo "current batched:"; RT_REGEN_FORM=batched timeout -k 10 120 python tools/ab_lib.py
