#!/bin/bash
# round-6 measurement batch (GPU box): headline + f64 bench lines, PMC traffic entries, SQ counter passes of the headline kernel,
# KL packed-FMA A/B, config 5 with RCCL in a world of one, ragged routing audit
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
export HIPNMF_SOURCE_COMMIT=${HIPNMF_SOURCE_COMMIT:-unknown} HIPNMF_ROUND=r06
cd $R
python3 bench.py --steps 3 --warmup 1 > $O/r06_bench.json 2> $O/bench.err; tail -c 600 $O/r06_bench.json; echo
python3 bench.py --dtype f64 --steps 3 --warmup 1 > $O/r06_bench_f64.json 2>> $O/bench.err
python3 bench.py --dtype f64 --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $O/r06_bench_f64_config4.json 2>> $O/bench.err
python3 bench.py --dtype f64 --config 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/r06_bench_f64_config2.json 2>> $O/bench.err
python3 bench.py --config 5 --force-nccl --steps 1 --warmup 1 --no-cpu-baseline > $O/r06_bench_config5_nccl.json 2>> $O/bench.err
for f in r06_bench_f64 r06_bench_f64_config4 r06_bench_f64_config2 r06_bench_config5_nccl; do python3 - <<PY
import json
d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["dtype"], d["roofline"]["frac"], d["parity"]["ok"], d.get("process_group_backend"), (d.get("collective") or {}).get("all_reduce_us_each_back_to_back"))
PY
done
# PMC traffic entries (two passes each)
bash tools/measure_traffic.sh > $O/traffic_headline.log 2>&1
bash tools/measure_traffic.sh --dtype f64 > $O/traffic_f64.log 2>&1
bash tools/measure_traffic.sh --m 128 --k 16 --batch 512 > $O/traffic_wide128.log 2>&1
bash tools/measure_traffic.sh --m 256 --k 16 --batch 256 --iters 100 > $O/traffic_wide256.log 2>&1
bash tools/measure_traffic.sh --m 64 --k 8 > $O/traffic_wide4.log 2>&1
cp $R/profiles/traffic.json $O/traffic.json
# SQ counters of the headline kernel on HEAD
PMC_KERNEL=fit_persistent bash tools/pmc_passes.sh r06_headline > $O/r06_pmc_fit_persistent_k5.txt 2>&1
PMC_KERNEL=fit_persistent bash tools/pmc_passes.sh r06_headline_f64 --dtype float64 > $O/r06_pmc_fit_persistent_f64_k5.txt 2>&1
# the compute ceiling with X served from L2 (all restarts of a trial share X)
python3 tools/shared_x_probe.py > $O/r06_shared_x_probe.log 2>&1
# KL narrow: packed FMAs (default build) vs scalar (-DHIPNMF_KL_NO_PK variant), same box
for lib in "$R/muscle_synergies_amd/lib/libhip_nmf.so" "$R/muscle_synergies_amd/lib/libhip_nmf_klnopk.so"; do
  for rep in 1 2; do HIPNMF_LIBRARY=$lib python3 tools/quick_bench.py --batch 2048 --iters 200 --threads 512 --loss kullback-leibler --rowmajor --reps 2 2>&1 | tail -1 | sed "s|^|KL narrow lib=${lib:-default(pk)} |"; done
done > $O/r06_kl_narrow_pk_ab.log 2>&1
cat $O/r06_kl_narrow_pk_ab.log
bash tools/routing_ragged_audit.sh $O > /dev/null 2>&1; tail -5 $O/routing_ragged.log
