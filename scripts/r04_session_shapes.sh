#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
for sh in 1280 9x12x12 3328 9x24x24; do
  timeout -k 10 600 python scripts/prof_forward_shapes.py $sh 8 2>&1 | grep -v amdgpu.ids > $O/r04_forward_shapes_$sh.log || exit 1
  cat $O/r04_forward_shapes_$sh.log
done
timeout -k 10 900 python scripts/run_configs.py 2>&1 | grep -v amdgpu.ids > $O/r04_configs_end_to_end.jsonl; cat $O/r04_configs_end_to_end.jsonl
