#!/bin/sh
# Builds the reference's own vendored samtools 1.3.1 (comparator / sambamba-shim back-end, SURVEY.md
# App. E) and its wgsim from the sources where they lie under /root/reference.  The tree is read-only and its
# Makefile builds in-tree, so the sources are staged in a scratch directory under /tmp; only the
# resulting binary lands in oracle/_ref/ (git-ignored, travels to the GPU box).  Test infrastructure.
set -e
SRC=/root/reference/src/samtools-1.3.1
OUT=$(cd "$(dirname "$0")" && pwd)/_ref
[ -d "$SRC" ] || { echo "no reference checkout: skipping"; exit 0; }
[ -x "$OUT/samtools" ] && [ -x "$OUT/wgsim" ] && exit 0
B=/tmp/ssg_samtools_build
rm -rf "$B" && mkdir -p "$B" && cp -r "$SRC"/. "$B"/ && chmod -R u+w "$B"
# ... and its read simulator, misc/wgsim (SURVEY.md 8d config 1 regenerates the reference's missing example FASTQ with it: misc/wgsim.c:438-463)
cd "$B" && ./configure --without-curses >/dev/null 2>&1 && make -j8 samtools misc/wgsim >/dev/null 2>&1
mkdir -p "$OUT" && cp "$B/samtools" "$OUT/samtools" && cp "$B/misc/wgsim" "$OUT/wgsim"
echo "built $OUT/samtools and $OUT/wgsim"
