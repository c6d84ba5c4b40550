#!/bin/bash
# SQ / LDS counter passes of the bench workload: tools/collect_pmc_extra.sh <tag>
set -e
tag=${1:-r02}
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="--plan-only --steps 6 --warmup 2"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $out/pmc_a -- python3 $root/bench.py $B > /dev/null 2> $out/pmc_a.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_b -- python3 $root/bench.py $B > /dev/null 2> $out/pmc_b.err
cd $root
python3 tools/pmc_summary.py $out/pmc_a 14 > $out/${tag}_pmc_sq_mfma_wait.txt
python3 tools/pmc_summary.py $out/pmc_b 14 > $out/${tag}_pmc_lds.txt
rm -rf $out/pmc_a $out/pmc_b
