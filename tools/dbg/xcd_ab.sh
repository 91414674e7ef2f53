B="python bench.py --no-cpu-baseline --no-states --no-app-run --steps 400"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['value']), round(d['ms_per_step'],4))"; }
export PCL_TUNE_FUSED_STEP=1
for i in 1 2; do
$B | pick plain
PCL_TUNE_XCD=3 $B | pick xcd
$B --state dense | pick plain_dense
PCL_TUNE_XCD=3 $B --state dense | pick xcd_dense
done
