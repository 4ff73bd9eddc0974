# float64, 9..32 channels, few long matrices: the lane-mapping kernels (HIPNMF_FORCE_WIDE=-1) vs everything on the matrix pipe (1) vs the library (0)
for m_k in "32 8" "24 7" "24 6" "32 4" "20 5" "16 5" "12 4" "16 8"; do
 for T_B in "3000 1" "10000 1" "10000 8" "30000 2" "100000 1" "1000000 1" "5000 40"; do
  set -- $m_k $T_B
  for v in -1 1 0; do
    printf 'float64 m=%d k=%d T=%d B=%d [FORCE_WIDE=%s] ' $1 $2 $3 $4 $v
    HIPNMF_FORCE_WIDE=$v python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 100 --threads 0 --rowmajor --dtype float64 2>&1 | tail -1 | awk '{print $6, $7, $NF}'
  done
 done
done
