B="python bench.py --no-cpu-baseline --no-states --no-app-run --steps 300"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['value']), round(d['ms_per_step'],4))"; }
timeout -k 10 600 python -m pytest tests/test_gpu_classic.py tests/test_gpu_apps.py tests/test_gpu_fuzz.py tests/test_gpu_fused_step.py -x -q -m gpu > gpurun_out/tf.log 2>&1; tail -2 gpurun_out/tf.log
$B | pick default &&
$B --state dense | pick dense &&
$B --state developed | pick developed &&
PCL_TUNE_ABLATE=1 $B | pick copy_only &&
PCL_TUNE_FUSED_STEP=0 $B | pick twopass
