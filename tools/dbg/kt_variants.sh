#!/bin/bash
# Diagnostic for DESIGN.md section 9: the table instance of the seeding kernel (ssg_k_smem_quad_kt, ssg_ktab.cpp) was wrong on the MI355X and
# right under emulation.  Variants of that ONE small translation unit, each with counters and an echo of the by-value arguments the kernel sees.
#   tools/dbg/kt_variants.sh build     here: libraries speedseq_amd/libssgpu_kt_*.so (git-ignored; they travel with gpurun)
#   tools/dbg/kt_variants.sh run       on the GPU box: intervals vs the oracle + counters for each
set -u
cd "$(dirname "$0")/../.."
variants="kt_base:-DSSG_KT_DBG kt_plain: kt_notab:-DSSG_KT_DBG,-DSSG_KT_NOTAB kt_check:-DSSG_KT_DBG,-DSSG_KT_CHECK kt_uniform:-DSSG_KT_DBG,-DSSG_KT_UNIFORM kt_uniform_plain:-DSSG_KT_UNIFORM kt_nopin:-DSSG_KT_DBG,-DSSG_NO_ASM_PINS kt_O1:-DSSG_KT_DBG,-O1 kt_t1:-DSSG_KT_DBG,-DSSG_SMQ_TRIPS=1 kt_w2:-DSSG_KT_DBG,-DSSG_SMQ_WAVES=2"
if [ "${1:-}" = build ]; then
  for v in $variants; do n=${v%%:*}; f=${v#*:}; make variant NAME=$n VUNITS=ssg_ktab VFLAGS="${f//,/ }" > /tmp/kt_build_$n.log 2>&1 && echo "built speedseq_amd/libssgpu_$n.so" || { echo "build of $n FAILED"; tail -5 /tmp/kt_build_$n.log; }; done
  exit 0
fi
out=gpurun_out; mkdir -p $out
for K in ${KT_KS:-8 0}; do
for lib in $(ls speedseq_amd/libssgpu_kt_*.so); do
  echo "== $lib  SSG_KTAB_K=$K"
  SSG_KTAB_K=$K SSG_KTAB_VERIFY=1 SSGPU_LIB=$PWD/$lib timeout 120 python tools/dbg/smem_dump.py ${KT_PAIRS:-500} ${KT_EMU:-} 2>&1 | tail -${KT_TAIL:-14}
done
done 2>&1 | tee $out/kt_variants.log
