#!/bin/bash
# Round-5 evidence for the large-M kernels (verdict r4 item 7 (ii)): where their cycles go.
#   rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES (+ a second pass with the MFMA / wave
#   counters) on the config-5-cut and config-4-cut steps (M = 1000 / 500: big_quad, big_accum / panel accumulate, gram_big)
# usage (GPU box, repo root): bash tools/pmc_big_kernels.sh   -> gpurun_out/r05/r05_big_kernels_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in 5cut 4cut; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $O/big_a_$cfg -o p -- python3 $R/tools/run_config.py $cfg 2 1 > $O/big_a_$cfg.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/big_b_$cfg -o p -- python3 $R/tools/run_config.py $cfg 2 1 > $O/big_b_$cfg.log 2>&1
done
cd $R
{
  for cfg in 5cut 4cut; do
    for p in a b; do
      echo "# config $cfg, pass $p"
      f=$(find $O/big_${p}_$cfg -name "*counter_collection.csv" | head -1)
      python3 tools/pmc_by_kernel.py $f | grep -E "kernel \||big_|gram_big|panel_mfma|prod_big|kept_wsum|col_wsum"
    done
  done
} > $O/r05_big_kernels_pmc.txt 2>&1
rm -rf $O/big_a_* $O/big_b_*
cat $O/r05_big_kernels_pmc.txt
