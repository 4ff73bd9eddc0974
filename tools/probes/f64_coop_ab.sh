# float64, 17..32 channels, k = 7 / 8, few long matrices: the lane-mapping kernels (HIPNMF_FORCE_WIDE=-1) vs the library's choice
for m_k in "32 8" "24 7"; do
 for T_B in "10000 1" "10000 8" "100000 1" "30000 2" "1000000 1" "5000 40"; do
  set -- $m_k $T_B
  for v in "HIPNMF_FORCE_WIDE=-1" "HIPNMF_FORCE_WIDE=0"; do
    printf 'float64 m=%d k=%d T=%d B=%d [%s] ' $1 $2 $3 $4 "$v"
    env $v python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 100 --threads 0 --rowmajor --dtype float64 2>&1 | tail -1 | awk '{print $6, $7, $NF}'
  done
 done
done
