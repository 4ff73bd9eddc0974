# float64 Frobenius, 9..16 channels, chip-filling batches of longer matrices: lane mappings (HIPNMF_FORCE_WIDE=-1) vs 4x4x4 (1) vs default (0)
for shape in "16 5 2500 2048" "16 5 10000 1024" "12 4 5000 2048" "10 3 2500 4096" "16 6 2500 2048" "14 5 1500 4096" "16 3 3000 2048"; do
  set -- $shape
  for v in "HIPNMF_FORCE_WIDE=-1" "HIPNMF_FORCE_WIDE=1" "HIPNMF_FORCE_WIDE=0"; do
    printf 'float64 m=%d k=%d T=%d B=%d [%s] ' $1 $2 $3 $4 "$v"
    env $v python tools/quick_bench.py --m $1 --k $2 --T $3 --batch $4 --iters 100 --threads 0 --rowmajor --dtype float64 2>&1 | tail -1 | awk '{print $6, $7, $9, $10, $NF}'
  done
done
