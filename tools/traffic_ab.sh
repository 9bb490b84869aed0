R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/traffic_ab
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPSA_ACCUM_PANEL=0
for lb in 0 16; do
  export GPSA_BIG_LB=$lb
  if [ $lb = 0 ]; then export GPSA_BA_NSPLIT=1; else export GPSA_BA_NSPLIT=4; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$lb -o f -- python3 $R/tools/bench_kernels.py big big5 > $O/f$lb.log 2>&1
  f=$(find $O/f$lb -name "*counter_collection.csv" | head -1)
  for k in big_quad big_accum gram_big_kernel; do echo "LB=$lb $k: $(python3 $R/tools/pmc_kernel.py $f $k | tr '\n' ' ')"; done
done
