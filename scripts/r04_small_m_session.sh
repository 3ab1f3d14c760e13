#!/bin/bash
# round-4: the DiT forward at small token counts with / without the split-K weight-streaming GEMM form (one box, same process order)
cd "$(dirname "$0")/.."
O=gpurun_out
AB=$PWD/mlx-video_amd/libltxk_ab.so
timeout -k 10 900 python -m pytest tests/test_gemm_epilogues_gpu.py tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -k "gemm" > $O/r04_gemm_tests2.log 2>&1
rc=$?; echo "pytest rc=$rc" >> $O/r04_gemm_tests2.log; tail -3 $O/r04_gemm_tests2.log
[ $rc -ne 0 ] && exit $rc
for v in "${@:-old:LTXK_GEMM_NT=4,LTXK_GEMM_KSPLIT=-1 new:}"; do
  name=${v%%:*}; envs=${v#*:}
  echo "==== $name ($envs)"
  env LTXK_LIB=$AB $(echo $envs | tr ',' ' ') timeout -k 10 600 python scripts/prof_small_m.py || exit 1
done > $O/r04_small_m_ab.log 2>&1
cat $O/r04_small_m_ab.log
