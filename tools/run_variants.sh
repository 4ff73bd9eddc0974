set -x
cd $GRAFT_REPO_ROOT
export HIPNMF_ROWLANE=0
python tools/quick_bench.py --batch 2048 --iters 200 --rowmajor --threads 512 --reps 2 2>&1 | grep -v amdgpu.ids
export HIPNMF_ROWLANE=1
for v in "" _rl_x0w0p1 _rl_x0w5p1 _rl_x1w5p1 _rl_x2w4p1 _rl_x0w0p2 _rl_x0w5p2; do
  echo "== variant $v"
  HIPNMF_LIBRARY=$GRAFT_REPO_ROOT/muscle_synergies_amd/lib/libhip_nmf$v.so python tools/quick_bench.py --batch 2048 --iters 200 --rowmajor --threads 512 --reps 2 2>&1 | grep -v amdgpu.ids
done
