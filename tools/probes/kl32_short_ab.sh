# fp32 Kullback-Leibler, 17..32 channels, k <= 5, short matrices: lane mappings vs 4x4x1 (HIPNMF_FORCE_WIDE=1)
for shape in "20 3 300 8192" "20 3 1000 4096" "32 5 300 8192" "32 5 1000 4096" "32 4 300 8192" "24 2 500 8192" "17 5 128 16384" "32 5 2500 2048"; do
  set -- $shape
  for v in "HIPNMF_FORCE_WIDE=-1" "HIPNMF_FORCE_WIDE=1"; do
    printf 'float32 m=%d k=%d T=%d B=%d [%s] ' $1 $2 $3 $4 "$v"
    env $v python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 100 --threads 0 --rowmajor --loss kullback-leibler 2>&1 | tail -1 | cut -c1-200
  done
done
