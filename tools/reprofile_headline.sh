tag=r03
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${tag}b
rm -rf $O && mkdir -p $O
run() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 "$@" > $O/$name.log 2>&1; f=$(find $O/$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_kernel_stats_$name.csv; }
run bench_steps2 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
run bench_config4 $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline
grep -h '^{' $O/bench_steps2.log > $O/${tag}_bench.json
grep -h '^{' $O/bench_config4.log > $O/${tag}_bench_config4.json
