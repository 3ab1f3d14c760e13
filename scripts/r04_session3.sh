#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 1100 python -m pytest tests/test_vae_stages_gpu.py tests/test_configs_gpu.py tests/test_sharding_gpu.py tests/test_golden.py tests/test_vae_gpu.py -x -q -m gpu > $O/r04_tests3.log 2>&1
rc=$?; echo "pytest rc=$rc" >> $O/r04_tests3.log; tail -4 $O/r04_tests3.log
[ $rc -ne 0 ] && exit $rc
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 scripts/pmc_families.py $O/pmc_r04_try $O/r04_pmc_try.json $O/r04_pmc_try.txt attn_1280x1280 norm_mod conv128 pixelnorm128 qknorm_k > $O/r04_pmc_try.log 2>&1
echo "pmc rc=$?"; cut -c1-400 $O/r04_pmc_try.log
