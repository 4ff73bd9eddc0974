# Kullback-Leibler at small batches: lane mappings (default below half the CUs) vs the 4x4 kernels (HIPNMF_FORCE_WIDE=1)
for shape in "float32 32 8 2500 1" "float32 32 8 10000 1" "float32 24 6 2500 16" "float32 32 8 200 16" "float64 32 8 2500 1" "float64 16 5 10000 1" "float64 24 6 1000 16" "float64 12 4 200 8"; do
  set -- $shape
  for v in "HIPNMF_FORCE_WIDE=-1" "HIPNMF_FORCE_WIDE=1"; do
    printf '%s m=%d k=%d T=%d B=%d [%s] ' $1 $2 $3 $4 $5 "$v"
    env $v python tools/quick_bench.py --m $2 --k $3 --T $4 --batch $5 --iters 200 --loss kullback-leibler --threads 0 --rowmajor --dtype $1 2>&1 | tail -1 | cut -c1-200
  done
done
