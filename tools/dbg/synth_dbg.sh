SSG_INDEX_VERBOSE=1 SSG_DEBUG=2 timeout 600 python bench.py --steps 1 --warmup 0 --no-e2e --cpu-sample 0 2>&1 | grep -E "round|seeds/read|ssg\] " | head -60
