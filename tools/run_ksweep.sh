cd $GRAFT_REPO_ROOT
for k in ${KS:-2 3 4 5 6 7 8}; do
  for rl in 0 1; do
    echo "== k=$k HIPNMF_ROWLANE=$rl"
    HIPNMF_ROWLANE=$rl python tools/quick_bench.py --batch 2048 --iters 200 --k $k --rowmajor --threads 512 --reps 3 2>&1 | grep "rep=2"
  done
done
