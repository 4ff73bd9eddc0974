#!/bin/bash
# Bytes the headline kernel moves beyond L2 per launch: two PMC passes over bench.py itself (program directly after
# `--`), then profiles/traffic.json keyed on the kernel instance and the workload.  Run on the GPU box:
#   gpurun -- 'HIPNMF_SOURCE_COMMIT=<git rev-parse --short HEAD> bash tools/measure_traffic.sh'   (optional: extra bench.py flags
#   after the script name, e.g. --m 64 --k 8 --batch 1024 for the wide-shape kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/traffic
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-resident --no-parity "$@" > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-resident --no-parity "$@" > $out/write.log 2>&1
python3 $R/tools/traffic_from_pmc.py $out "$@"
