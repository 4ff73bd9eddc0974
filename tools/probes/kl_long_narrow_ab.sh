# Kullback-Leibler, one long narrow matrix: the lane mappings' one workgroup (HIPNMF_KL_SLICED=0) vs the library's choice
for dt in float32 float64; do
 for m_k in "4 2" "8 4" "12 4" "16 5" "16 8" "24 6"; do
  for T_B in "1500 1" "2500 1" "5000 1" "10000 1" "2500 8" "5000 32"; do
  set -- $m_k $T_B
  for v in 0 1; do
    printf '%s m=%d k=%d T=%d B=%d [KL_SLICED=%s] ' $dt $1 $2 $3 $4 $v
    HIPNMF_KL_SLICED=$v python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 100 --threads 0 --rowmajor --dtype $dt --loss kullback-leibler 2>&1 | tail -1 | awk '{print $6, $7, $NF}'
  done
  done
 done
done
