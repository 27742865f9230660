run() { python bench.py --steps 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],1), d['config']['index_build_s'], {x:k[x] for x in k if 'sal' in x})"; }
SSG_SA_INTV=4 run intv4
SSG_SA_INTV=2 run intv2
SSG_SA_INTV=1 run intv1
