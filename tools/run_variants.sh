# usage: bash tools/run_variants.sh <threads> <ROWLANE 0|1> <variant-suffix|default>...   (libraries built with
# python -m muscle_synergies_amd.build --variant <suffix> --only <tu> --flag=...)
cd $GRAFT_REPO_ROOT
threads=$1; shift
export HIPNMF_ROWLANE=$1; shift
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/muscle_synergies_amd/lib/libhip_nmf_$v.so
  [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/muscle_synergies_amd/lib/libhip_nmf.so
  echo "== variant $v threads=$threads rowlane=$HIPNMF_ROWLANE"
  HIPNMF_LIBRARY=$lib python tools/quick_bench.py --batch ${BATCH:-2048} --iters 200 --rowmajor --threads $threads --reps 3 2>&1 | grep "rep=[12]"
done
