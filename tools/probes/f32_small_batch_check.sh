# the library's choice after the routing change (compare with tools/probes/f32_small_batch_ab.sh's two columns)
for m_k in "32 8" "24 6" "32 4" "28 7"; do
 for T_B in "600 8" "3000 1" "3000 32" "3000 100" "10000 1" "10000 100" "30000 2" "100000 1" "300 60"; do
  set -- $m_k $T_B
  for v in "HIPNMF_FORCE_WIDE=-1" "HIPNMF_FORCE_WIDE=0"; do
    printf 'float32 m=%d k=%d T=%d B=%d [%s] ' $1 $2 $3 $4 "$v"
    env $v python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 200 --threads 0 --rowmajor 2>&1 | tail -1 | awk '{print $6, $7, $NF}'
  done
 done
done
