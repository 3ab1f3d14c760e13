#!/bin/bash
# PMC passes over one attention shape: scripts/pmc_attn.sh OUTDIR Tq Tk [LABEL...]   (one set of passes per label; the label
# only names the output directory - build variants are selected with LTXK_LIB=<other build of libltxk>)
out=$1; tq=$2; tk=$3; shift 3
export TMPDIR=/tmp
mkdir -p $out
[ $# -eq 0 ] && set -- base
for v in "$@"; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/v${v}_a -- python scripts/prof_attn_rand.py $tq $tk 6 > $out/v${v}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $out/v${v}_b -- python scripts/prof_attn_rand.py $tq $tk 6 > $out/v${v}_b.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/v${v}_t -- python scripts/prof_attn_rand.py $tq $tk 6 > $out/v${v}_t.log 2>&1
done
python scripts/pmc_summary.py $out
