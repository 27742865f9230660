#!/bin/bash
# Diagnostic for DESIGN.md section 9: the table instance of the seeding kernel (ssg_k_smem_quad_kt, ssg_ktab.cpp) was wrong on the MI355X and
# right under emulation.  Variants of that ONE small translation unit, each with counters and an echo of the by-value arguments the kernel sees.
#   tools/dbg/kt_variants.sh build     here: libraries speedseq_amd/libssgpu_kt_*.so (git-ignored; they travel with gpurun)
#   tools/dbg/kt_variants.sh run       on the GPU box: intervals vs the oracle + counters for each
set -u
cd "$(dirname "$0")/../.."
# round 4, first call: base / plain / notab / check / nopin / O1 / t1 / w2 all wrong (234 of 1000 reads), uniform and uniform_plain right (profiles/r04a_kt_variants.log);
# second call: which of the three parts of "uniform" it takes
variants="${KT_VARIANTS:-kt_base:-DSSG_KT_DBG kt_plain: kt_uniform_plain:-DSSG_KT_UNIFORM kt_exit:-DSSG_KT_U_EXIT kt_site:-DSSG_KT_U_SITE kt_inner:-DSSG_KT_U_INNER kt_exit_site:-DSSG_KT_U_EXIT,-DSSG_KT_U_SITE kt_exit_inner:-DSSG_KT_U_EXIT,-DSSG_KT_U_INNER kt_site_inner:-DSSG_KT_U_SITE,-DSSG_KT_U_INNER}"
if [ "${1:-}" = build ]; then
  for v in $variants; do n=${v%%:*}; f=${v#*:}; make variant NAME=$n VUNITS=ssg_ktab VFLAGS="${f//,/ }" > /tmp/kt_build_$n.log 2>&1 && echo "built speedseq_amd/libssgpu_$n.so" || { echo "build of $n FAILED"; tail -5 /tmp/kt_build_$n.log; }; done
  exit 0
fi
out=gpurun_out; mkdir -p $out
for K in ${KT_KS:-8 0}; do
for lib in $(ls speedseq_amd/libssgpu_kt_*.so); do
  echo "== $lib  SSG_KTAB_K=$K"
  SSG_KTAB_K=$K SSG_KTAB_VERIFY=1 SSGPU_LIB=$PWD/$lib timeout 120 python tools/dbg/smem_dump.py ${KT_PAIRS:-500} ${KT_EMU:-} 2>&1 | tail -${KT_TAIL:-24}
done
done 2>&1 | tee $out/kt_variants.log
