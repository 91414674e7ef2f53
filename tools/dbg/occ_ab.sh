B="python bench.py --no-cpu-baseline --no-states --no-app-run --steps 400"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['value']), round(d['ms_per_step'],4), {k[-8:]:round(v,4) for k,v in d['roofline']['avg_ms'].items() if v})"; }
export PCL_TUNE_FUSED_STEP=0
for i in 1 2; do
for L in occ1 occ4; do
PCL_LIB_OVERRIDE=build/libs/libpyclaw_amd_$L.so $B | pick $L
PCL_LIB_OVERRIDE=build/libs/libpyclaw_amd_$L.so $B --state dense | pick ${L}_dense
done; done
