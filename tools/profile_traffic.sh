#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE; separate --pmc passes) of the large-M kernels at BASELINE config 4 | 5.
# usage (repo root, GPU box): bash tools/profile_traffic.sh 4|5
set -u
W=${1:-5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/traffic_c$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/tools/run_config.py $W 1 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/tools/run_config.py $W 1 1 > $O/write.log 2>&1
cd $R
for k in big_quad big_accum gram_big_kernel col_wsum_rows; do
  for c in fetch write; do
    f=$(find $O/$c -name "*counter_collection.csv" | head -1)
    echo "$k $c: $(python3 tools/pmc_kernel.py $f $k | tr '\n' ' ')"
  done
done
