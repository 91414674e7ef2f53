#!/bin/bash
# Several ranks of tests/mp_gpu_worker.py on ONE GPU (host-staged halo wire).  usage: tools/run_mp.sh case nranks [overlap]
P=$((20000 + RANDOM % 20000))
export MASTER_ADDR=127.0.0.1 MASTER_PORT=$P TORCHELASTIC_RUN_ID=dbg$P PCL_HALO_TRANSPORT=host PCL_FORCE_DEVICE=0 WORLD_SIZE=$2
if [ -n "$3" ]; then export PCL_HALO_OVERLAP=$3; fi
mkdir -p gpurun_out
for r in $(seq 0 $(($2-1))); do
  RANK=$r LOCAL_RANK=$r timeout 120 python tests/mp_gpu_worker.py $1 > gpurun_out/mp_$1_$r.log 2>&1 &
done
wait
tail -n 3 gpurun_out/mp_$1_0.log
