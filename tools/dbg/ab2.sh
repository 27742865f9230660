run() { python bench.py --steps 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],1), {x:k[x] for x in k if 'smem' in x})"; }
timeout 100 python -m pytest tests -m gpu -x -q -k "smem or align1_150" 2>&1 | tail -1
SSG_SMEM_LPR=1 timeout 100 python -m pytest tests -m gpu -x -q -k "smem or align1_150 or pe_sam_150" 2>&1 | tail -1
run lpr4
SSG_SMEM_LPR=1 run lpr1_w16
SSG_SMEM_LPR=1 SSG_SMEM_WAVES_PER_CU=8 run lpr1_w8
