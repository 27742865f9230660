# A/B helper: run the bench with alternative library builds (SSGPU_LIB) and print throughput + the kernels matching $PAT
run() { python bench.py --steps 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],1), {x:k[x] for x in k if os.environ.get('PAT','smem') in x})"; }
for v in "$@"; do if [ "$v" = default ]; then run default; else SSGPU_LIB=$PWD/speedseq_amd/libssgpu_$v.so run $v; fi; done
