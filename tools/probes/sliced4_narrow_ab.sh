# up to 32 channels, row-sliced general path: 16x16x4 (default) vs the 4x4 kernels (HIPNMF_WIDE4_SLICED_NARROW=1)
for shape in "float64 32 8 3000 32" "float64 32 8 10000 1" "float64 24 7 30000 2" "float64 32 8 10000 100" "float64 24 6 3000 32" "float64 32 3 3000 32" "float32 32 8 3000 32" "float32 32 8 10000 100" "float32 24 6 3000 32" "float32 32 8 10000 1"; do
  set -- $shape
  for v in 0 1; do
    printf '%s m=%d k=%d T=%d B=%d [narrow4=%s] ' $1 $2 $3 $4 $5 $v
    HIPNMF_WIDE4_SLICED_NARROW=$v python tools/quick_bench.py --m $2 --k $3 --T $4 --batch $5 --iters 200 --threads 0 --rowmajor --dtype $1 2>&1 | tail -1 | awk '{print $6, $7, $NF}'
  done
done
