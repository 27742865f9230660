#!/bin/bash
# chain stage A/B on the GPU box: stage wall times (SSG_DEBUG=1 syncs after every stage) and kernel times under the chaining switches, then the chaining tests
tag=${1:-chain_ab}
out=gpurun_out; mkdir -p $out
SSG_DEBUG=1 timeout 900 python tools/smem_ab.py --steps 2 --kernels chain --out $out/${tag}.json base lds0:SSG_CHAIN_LDS=0 ws0:SSG_CHAIN_WSORT=0 both0:SSG_CHAIN_LDS=0,SSG_CHAIN_WSORT=0 > $out/${tag}.log 2>&1
grep -E "stage (sal|chain) |\"config\"|summary counts" $out/${tag}.log | cut -c1-700 | tail -60
timeout 900 python -m pytest tests -m gpu -x -q -k "chain or repeats or light" > $out/${tag}_pytest.log 2>&1; tail -5 $out/${tag}_pytest.log
