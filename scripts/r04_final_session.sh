#!/bin/bash
# round-4 measurement record on one box: full GPU test suite (parity ledger), PMC passes of every kernel family on the final
# sources, bench.py, kernel traces of bench.py and of the VAE decode.  Everything lands under gpurun_out/.
cd "$(dirname "$0")/.."
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04_tests_final.log 2>&1
rc=$?; echo "pytest rc=$rc" >> $O/r04_tests_final.log; tail -3 $O/r04_tests_final.log
[ $rc -ne 0 ] && exit $rc
cp $O/parity_measured.json $O/r04_parity.json
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf $O/pmc_r04
timeout -k 10 1500 python3 scripts/pmc_families.py $O/pmc_r04 $O/r04_pmc_traffic.json $O/r04_pmc_summary.txt > $O/r04_pmc.log 2>&1
echo "pmc rc=$?"; tail -2 $O/r04_pmc.log | cut -c1-200
mkdir -p profiles && cp $O/r04_pmc_traffic.json profiles/r04_pmc_traffic.json      # bench.py quotes it (source_sha-matched) in this same call
python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err; echo "bench rc=$?"
export LTXK_BENCH_EXTRAS=0
rm -rf $O/prof_bench_r04
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench_r04 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r04_bench_under_rocprof.json 2> $O/r04_bench_under_rocprof.err
cp $(find $O/prof_bench_r04 -name "*kernel_stats.csv" | head -1) $O/r04_final_kernel_stats.csv
rm -rf $O/prof_vae_r04
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vae_r04 -- python3 scripts/prof_vae_decode.py 11 > $O/r04_vae_decode.log 2>&1
cp $(find $O/prof_vae_r04 -name "*kernel_stats.csv" | head -1) $O/r04_vae_decode_kernel_stats.csv
python3 scripts/trace_summary.py $O/prof_bench_r04 144 | head -24
