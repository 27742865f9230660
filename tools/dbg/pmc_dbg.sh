cd /tmp && export TMPDIR=/tmp
export SSG_INDEX_VERBOSE=1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_dbg -o pmc -- python -X faulthandler /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-e2e --no-profile 2>&1 | grep -v "^    @\|bucket" | grep -A25 -i "fault\|Thread\|File" | head -60
