for i in 1 2; do
for p in 1 0; do
GPSA_ELBO_PAIR=$p python bench.py --headline-only --blocks 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('PAIR=$p', round(d['ms_per_step'],4), 'elbo', round(d['roofline']['avg_launch_ms'],4), 'gram', round(d['roofline']['other_kernels']['quadform_bwd_omega']['avg_ms'],4))"
done
done
