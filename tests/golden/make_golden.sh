#!/bin/sh
# Regenerates tests/golden/ from the reference checkout (run in the build container only;
# /root/reference does not exist on the GPU box).  The FM-index files are the reference's own
# bundled `bwa index` output and are the golden vectors pinning the oracle's and the product's
# index format.
set -e
SRC=/root/reference/example/data/human_g1k_v37_20_42220611-42542245.fasta
DST=$(dirname "$0")
cp "$SRC" "$DST/chr20_slice.fa"
for e in amb ann bwt pac sa; do cp "$SRC.$e" "$DST/chr20_slice.fa.$e"; done
chmod u+w "$DST"/chr20_slice.fa*
# The reference's own driver script, held as a fixture so that the `-m gpu` test can run the UNMODIFIED `speedseq align`
# (bin/speedseq:189-504) on the product executables on the GPU box, where /root/reference does not exist.
cp /root/reference/bin/speedseq "$DST/speedseq_ref_script.sh"
chmod u+w "$DST/speedseq_ref_script.sh"
