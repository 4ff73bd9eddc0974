mkdir -p gpurun_out/r06
echo "=== exit backtrace probe: ordinary launch (default)"; bash tools/probes/exit_bt.sh 2>&1 | grep -E "^==|rc=" 
echo "=== exit backtrace probe: HIPNMF_COOP_LAUNCH=1 (cooperative launch API)"; HIPNMF_COOP_LAUNCH=1 bash tools/probes/exit_bt.sh 2>&1 | grep -E "^==|rc=|libhsa|libamdhip" | head -30
python tools/exit_cases.py --reps 8 --log gpurun_out/r06/exit_cases_fixed.log > /dev/null 2>&1; tail -16 gpurun_out/r06/exit_cases_fixed.log
echo "=== config 2 A/B"
for v in 0 1; do for rep in 1 2; do HIPNMF_COOP_LAUNCH=$v python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('COOP_LAUNCH=$v', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms_avg'], d['parity']['ok'])"; done; done
python -X faulthandler -m pytest tests/test_gpu_exit.py tests/test_gpu_tsharded_devices.py tests/test_gpu_pipeline.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15
