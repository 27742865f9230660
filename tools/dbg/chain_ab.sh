#!/bin/bash
# chain stage A/B on the GPU box: stage wall times (SSG_DEBUG=1 syncs after every stage) and kernel times under the chaining switches, the phase
# counters of the wave kernels from the tuning build, then the chaining tests
tag=${1:-chain_ab}
out=gpurun_out; mkdir -p $out
SSG_DEBUG=1 timeout 900 python tools/smem_ab.py --steps 2 --kernels chain --dbg-cycles --out $out/${tag}.json base ws0:SSG_CHAIN_WSORT=0 both0:SSG_CHAIN_LDS=0,SSG_CHAIN_WSORT=0 t@tune > $out/${tag}.log 2>&1
grep -E "stage (sal|chain) |\"config\"|summary counts" $out/${tag}.log | cut -c1-900 | tail -60
timeout 900 python -m pytest tests -m gpu -x -q -k "chain or repeats or light" > $out/${tag}_pytest.log 2>&1; tail -5 $out/${tag}_pytest.log
