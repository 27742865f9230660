run() { python bench.py --steps 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],1), d['config']['index_build_s'], {x:k[x] for x in k if 'smem' in x})"; }
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
run k12
SSG_KTAB_K=13 run k13
SSG_KTAB_K=0 run k0
