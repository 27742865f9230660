#!/bin/bash
# Compile-time variants of the seeding unit (ssg_seed.cpp) next to the product build: speedseq_amd/libssgpu_seed_<name>.so (git-ignored; they
# travel with gpurun).  tools/smem_ab.py times them inside the bench's step:  smem_ab.py base w2@seed_w2 ...
set -u
cd "$(dirname "$0")/../.."
variants="${SEED_VARIANTS:-seed_w2:-DSSG_S2_WAVES=2 seed_w3:-DSSG_S2_WAVES=3 seed_t1:-DSSG_S2_TRIPS=1 seed_t3:-DSSG_S2_TRIPS=3}"
for v in $variants; do n=${v%%:*}; f=${v#*:}; make variant NAME=$n VUNITS=ssg_seed VFLAGS="${f//,/ }" > /tmp/seed_build_$n.log 2>&1 && echo "built speedseq_amd/libssgpu_$n.so" || { echo "build of $n FAILED"; tail -5 /tmp/seed_build_$n.log; }; done
