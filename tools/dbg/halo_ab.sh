#!/bin/bash
# A/B of the halo stream priority at the C4 block size
for P in 1 0 1 0; do
  echo "PCL_HALO_PRIORITY=$P"
  PCL_HALO_PRIORITY=$P python tools/halo_overlap_bench.py 4096 2048 300 2>&1 | grep ms/step
done
