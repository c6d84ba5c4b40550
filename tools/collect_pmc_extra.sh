set -e
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $out/pmc_a -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-fwd-only > /dev/null 2> $out/pmc_a.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_b -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-fwd-only > /dev/null 2> $out/pmc_b.err
cd $root
python3 tools/pmc_summary.py $out/pmc_a 14 > $out/r01_pmc_sq_mfma_wait.txt
python3 tools/pmc_summary.py $out/pmc_b 14 > $out/r01_pmc_lds.txt
rm -rf $out/pmc_a $out/pmc_b
