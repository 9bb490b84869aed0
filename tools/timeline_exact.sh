R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tlx
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPSA_EXACT_GRAD=1
rocprofv3 --kernel-trace --output-format csv -d $O/k1 -o t -- python3 $R/bench.py --no-cpu-baseline --no-graph --no-s1 --steps 6 --warmup 3 > $O/k1.log 2>&1
cd $R
f=$(find $O/k1 -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $f > $O/k1_exact.timeline.txt
tail -1 $O/k1_exact.timeline.txt
rm -rf $O/k1
