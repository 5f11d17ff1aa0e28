# The mixed form's launch parameters at 1920x1080 through their environment overrides, twice each:
#   bash tools/sweep_mixed.sh
# (waves sparser than 8 lanes need a library whose clamp of RT_SPARSE_SHIFT in launch_mega_t is raised from 3 to 6)
for rep in 1 2; do
  for div in 32 64 128 256 512; do
    echo -n "chain_div $div: "; RT_REGEN_CHAIN_DIV=$div timeout -k 10 100 python3 tools/ab_lib.py 2>&1 | grep -v amdgpu
  done
  for sh in 1 2 3 4 5 6; do
    for sp in 0 1 2 3 4 8; do
      echo -n "sparse_shift $sh sparse_ranks $sp: "; RT_SPARSE_SHIFT=$sh RT_SPARSE_RANKS=$sp timeout -k 10 100 python3 tools/ab_lib.py 2>&1 | grep -v amdgpu
    done
  done
done
