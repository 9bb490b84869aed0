#!/bin/bash
# Round-6 evidence for the fp64 kernels' "T = T(MFMA loop) + bytes / 8.7 TB/s" law (verdict r5 item 4a): the memory-path
# counters of proj64_kernel / longk64_kernel / whiten_mfma_kernel inside the headline step, next to the two contraction
# kernels for scale.  Separate --pmc passes (a pass holds what fits the hardware's counter slots); no tracing options.
# usage (GPU box, repo root): bash tools/pmc_fp64.sh   -> gpurun_out/r06/r06_fp64_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --headline-only --blocks 1 --steps 3 --warmup 1"
rocprofv3 -L > $O/counters_avail.txt 2>&1
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
P3="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
P4="TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"
P5="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
P6="TCC_EA_WRREQ_STALL_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_BUSY_avr"
P7="TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_64B_sum"
P8="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7" "$P8"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $O/f64_$i -o p -- python3 $B > $O/f64_$i.log 2>&1
done
cd $R
{
  echo "# headline step (bench.py --headline-only), per-kernel MEANS of each counter per launch; one rocprofv3 --pmc pass per block"
  for i in 1 2 3 4 5 6 7 8; do
    f=$(find $O/f64_$i -name "*counter_collection.csv" | head -1)
    echo "## pass $i"
    if [ -n "$f" ]; then
      python3 tools/pmc_by_kernel.py $f | grep -E "kernel \||proj64|longk64_kernel|whiten_mfma|panel_elbo|gram_mfma|kmat_fwd|kmat_bwd_d2"
    else
      echo "(no counter file: $(tail -2 $O/f64_$i.log | tr '\n' ' '))"
    fi
  done
} > $O/r06_fp64_pmc.txt 2>&1
rm -rf $O/f64_*/
cat $O/r06_fp64_pmc.txt
