#!/bin/bash
# rocprofv3 kernel statistics of the round's benchmarks (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r03')
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
rm -rf $O && mkdir -p $O
run() {  # name, program args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats_$name.csv
}
run bench_steps2 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
run bench_wide_m64_k8 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --m 64 --k 8
run bench_wide_m128_k16 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --m 128 --k 16 --batch 512
run bench_config4 $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline
run bench_config5 $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline
run bench_config2 $R/bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline
run filter_bench $R/tools/filter_bench.py --orders 4 --dtypes float32
run envelope_bench $R/tools/envelope_bench.py
for n in bench_steps2 bench_wide_m64_k8 bench_wide_m128_k16 bench_config4 bench_config5 bench_config2; do
  grep -h '^{' $O/$n.log > $O/${tag}_${n/bench_steps2/bench}.json
done
ls $O/*.csv $O/*.json
