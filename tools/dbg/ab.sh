run() { python bench.py --steps 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],1), {x:k[x] for x in k if 'smem' in x})"; }
timeout 120 python -m pytest tests -m gpu -x -q -k "smem or align1_150 or pe_sam_150" 2>&1 | tail -2
SSG_SMEM_KERNEL=lane run lane
SSGPU_LIB=$PWD/speedseq_amd/libssgpu_q4.so SSG_SMEM_WAVES_PER_CU=16 run q4w16
SSGPU_LIB=$PWD/speedseq_amd/libssgpu_q6.so SSG_SMEM_WAVES_PER_CU=24 run q6w24
SSGPU_LIB=$PWD/speedseq_amd/libssgpu_q8.so SSG_SMEM_WAVES_PER_CU=32 run q8w32
