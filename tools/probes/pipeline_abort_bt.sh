#!/bin/bash
# native backtrace of the abort at the first pageable copy after the pipelined tests (1 of 22 full-suite runs after the HostBatch fix)
R=$(cd "$(dirname "$0")/../.." && pwd)
gcc -shared -fPIC -O1 -o /tmp/segv_bt.so $R/tools/probes/segv_bt.c || exit 1
cd $R
for i in $(seq 1 ${1:-14}); do
  LD_PRELOAD=/tmp/segv_bt.so timeout 300 python3 -m pytest tests/test_gpu_multi_device.py tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -q -x -p no:cacheprovider -p no:faulthandler > /tmp/pa_$i.log 2>&1
  rc=$?
  echo "rep $i rc=$rc $(tail -1 /tmp/pa_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -v "amdgpu.ids\|/usr/local/bin/python" /tmp/pa_$i.log | tail -40; fi
done
