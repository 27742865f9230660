#!/bin/bash
# chain stage A/B on the GPU box: stage wall times (SSG_DEBUG=1 syncs after every stage) and kernel times under the chaining switches given as
# smem_ab.py configurations (default: the shipped form against each switch turned off), then the chaining tests
tag=${1:-chain_ab}; shift
cfgs=${@:-base spec0:SSG_CHAIN_SPEC=0 ws0:SSG_CHAIN_WSORT=0 lds0:SSG_CHAIN_LDS=0}
out=gpurun_out; mkdir -p $out
SSG_DEBUG=1 timeout 900 python tools/smem_ab.py --steps 2 --kernels ${KERNELS:-chain} --dbg-cycles --out $out/${tag}.json $cfgs > $out/${tag}.log 2>&1
grep -E "stage (${STAGES:-sal|chain}) |\"config\"|summary counts" $out/${tag}.log | cut -c1-900 | tail -60
timeout 900 python -m pytest tests -m gpu -x -q -k "chain or repeats or light" > $out/${tag}_pytest.log 2>&1; tail -5 $out/${tag}_pytest.log
