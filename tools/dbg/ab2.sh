run() { python bench.py --steps 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],1), {x:k[x] for x in k if 'smem_q' in x or 'matesw' in x})"; }
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
run default
SSGPU_LIB=$PWD/speedseq_amd/libssgpu_q5.so SSG_SMEM_WAVES_PER_CU=20 run q5w20
SSG_SMEM_LPR=4 run lpr4
