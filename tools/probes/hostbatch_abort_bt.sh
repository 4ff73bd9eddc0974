#!/bin/bash
# native backtrace of the abort seen in tests/test_gpu_pipeline.py::test_host_batch_* (3 of 8 full-suite runs, round 6)
R=$(cd "$(dirname "$0")/../.." && pwd)
gcc -shared -fPIC -O1 -o /tmp/segv_bt.so $R/tools/probes/segv_bt.c || exit 1
cd $R
for i in $(seq 1 ${1:-12}); do
  LD_PRELOAD=/tmp/segv_bt.so python3 -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -p no:cacheprovider -p no:faulthandler > /tmp/hb_$i.log 2>&1
  rc=$?
  echo "rep $i rc=$rc $(tail -1 /tmp/hb_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -v amdgpu.ids /tmp/hb_$i.log | tail -45; fi
done
