# few long matrices, Kullback-Leibler: one workgroup per matrix (HIPNMF_FORCE_BIG=0) vs the row-sliced one-pass general-shape kernel (1): where is the boundary
for dt in float32 float64; do
 for m_k in "32 8" "64 8" "128 6" "16 5"; do
  for T_B in "1000 1" "2500 1" "5000 1" "2500 8" "1000 32" "2500 32" "5000 32" "2500 64" "5000 64" "2500 128" "10000 128" "5000 256"; do
  set -- $m_k $T_B
  for v in 0 1; do
    printf '%s m=%d k=%d T=%d B=%d [FORCE_BIG=%s] ' $dt $1 $2 $3 $4 $v
    HIPNMF_FORCE_WIDE=1 HIPNMF_FORCE_BIG=$v python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 100 --threads 0 --rowmajor --dtype $dt --loss kullback-leibler 2>&1 | tail -1 | awk '{print $6, $7, $NF}'
  done
  done
 done
done
