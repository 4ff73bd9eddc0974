# Kullback-Leibler on the 4x4 kernels: 4 waves per workgroup (round 4) vs 8
for shape in "float32 64 8 2500 4096" "float32 64 8 10000 1024" "float32 48 6 2500 4096" "float32 64 4 2500 4096" "float32 128 8 1000 2048" "float32 96 4 1000 2048" "float32 32 8 2500 4096" "float32 32 8 300 8192" "float32 24 6 600 100" \
             "float64 64 8 2500 2048" "float64 48 6 2500 2048" "float64 64 4 2500 2048" "float64 32 8 2500 2048" "float64 16 5 2500 2048" "float64 32 8 128 8192" "float64 24 6 1000 16" "float64 32 8 2500 1"; do
  set -- $shape
  for v in 4 8; do
    printf '%s m=%d k=%d T=%d B=%d [waves=%s] ' $1 $2 $3 $4 $5 $v
    HIPNMF_KL_WAVES=$v python tools/quick_bench.py --m $2 --k $3 --T $4 --batch $5 --iters 100 --threads 0 --rowmajor --dtype $1 --loss kullback-leibler 2>&1 | tail -1 | awk '{print $6, $7, $9, $10, $NF}'
  done
done
