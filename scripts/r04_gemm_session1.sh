#!/bin/bash
# round-4 GPU session 1: correctness of the 128-column tile / split-K forms, then their timings (one box, interleaved)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out
AB=$PWD/mlx-video_amd/libltxk_ab.so
timeout -k 10 900 python -m pytest tests/test_gemm_epilogues_gpu.py tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -k "gemm" > $O/r04_gemm_tests1.log 2>&1
rc=$?; echo "pytest rc=$rc" >> $O/r04_gemm_tests1.log; tail -5 $O/r04_gemm_tests1.log
[ $rc -ne 0 ] && exit $rc
SH="1280:4096:4096:0 1280:4096:4096:3 1280:4096:4096:4 1280:4096:16384:3 1280:8192:4096:0 1280:16384:4096:1 1296:4096:4096:3 1296:4096:16384:3 3328:4096:4096:3 3328:4096:16384:3 5184:4096:4096:3 5184:4096:16384:3 1024:8192:4096:0:2 2560:4096:4096:3"
timeout -k 10 1500 python scripts/ab_gemm.py 3 "$SH" "nt4:LTXK_LIB=$AB,LTXK_GEMM_NT=4,LTXK_GEMM_KSPLIT=-1" "nt2:LTXK_LIB=$AB,LTXK_GEMM_NT=2,LTXK_GEMM_KSPLIT=-1" "auto:LTXK_LIB=$AB" > $O/r04_gemm_128col_ab.log 2>&1
cat $O/r04_gemm_128col_ab.log
