#!/bin/bash
# Diagnostic for DESIGN.md section 9's open item: variants of the seeding kernel that execute the same statements but differ in kernarg
# layout / register allocation gave wrong intervals on the MI355X (round 3) while the host emulation agreed with the oracle.
#   tools/dbg/smem_variants.sh build     here (CPU): A/B libraries next to the product one (git-ignored, they travel with gpurun)
#   tools/dbg/smem_variants.sh run       on the GPU box: the seeding and end-to-end kernel tests against the oracle with each of them
# probe   = two unused trailing kernel arguments (the layout of the broken round-3 builds)      probe_O1 = the same at -O1
# probe_nopin / base_nopin = without the inline-asm register pins of ssg_bwt_extend1_lean (the only statements of the seeding path the emulation does not execute)
# probe_w2 / base_w2 = 2 waves per SIMD (256 VGPRs available) with / without the extra arguments; probe_t1 = one state-machine trip per round
set -u
cd "$(dirname "$0")/../.."
variants="probe:-DSSG_SMQ_PROBE probe_O1:-DSSG_SMQ_PROBE,-O1 probe_w2:-DSSG_SMQ_PROBE,-DSSG_SMQ_WAVES=2 base_w2:-DSSG_SMQ_WAVES=2 probe_t1:-DSSG_SMQ_PROBE,-DSSG_SMQ_TRIPS=1 probe_nopin:-DSSG_SMQ_PROBE,-DSSG_NO_ASM_PINS base_nopin:-DSSG_NO_ASM_PINS"
if [ "${1:-}" = build ]; then
  for v in $variants; do n=${v%%:*}; f=${v#*:}; make variant NAME=$n VFLAGS="${f//,/ }" > /dev/null 2>&1 && echo "built speedseq_amd/libssgpu_$n.so" || echo "build of $n FAILED"; done
  exit 0
fi
out=gpurun_out; mkdir -p $out
for lib in speedseq_amd/libssgpu.so $(ls speedseq_amd/libssgpu_probe*.so speedseq_amd/libssgpu_base_*.so 2>/dev/null); do
  echo "== $lib"
  SSGPU_LIB=$PWD/$lib timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "smem or pe_sam" 2>&1 | tail -3
  SSGPU_LIB=$PWD/$lib timeout 120 python tools/dbg/smem_dump.py 500 2>&1 | tail -16    # which intervals are extra / missing, for the first reads that differ
done 2>&1 | tee $out/smem_variants.log
