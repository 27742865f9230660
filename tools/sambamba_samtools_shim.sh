#!/bin/sh
# TEST INFRASTRUCTURE (comparator for bin/sambamba, tests/test_speedseq_script.py): sambamba's command line as the
# reference issues it (bin/speedseq:440-448,491-495) mapped onto the reference's vendored samtools 1.3.1
# built into oracle/_ref/ (or env SSG_SAMTOOLS).  The product's sambamba is speedseq_amd/host/sambamba_main.cpp.
HERE=$(cd "$(dirname "$0")" && pwd)
ST=${SSG_SAMTOOLS:-$HERE/../oracle/_ref/samtools}
[ -x "$ST" ] || { echo "sambamba shim: samtools not found ($ST); set SSG_SAMTOOLS" >&2; exit 127; }
cmd=$1; shift
case "$cmd" in
view)
	lvl=""; hdr=0; in=""
	while [ $# -gt 0 ]; do
		case "$1" in
		-S) ;; -f) shift ;; -l) shift; lvl=$1 ;; -H) hdr=1 ;; -t) shift ;; *) in=$1 ;;
		esac; shift
	done
	[ "$in" = "/dev/stdin" ] && in=-
	if [ $hdr -eq 1 ]; then exec "$ST" view -H "$in"; fi
	if [ "$lvl" = "0" ]; then exec "$ST" view -b -u "$in"; else exec "$ST" view -b "$in"; fi ;;
sort)
	t=1; m=1G; tmp=.; out=""; in=""
	while [ $# -gt 0 ]; do
		case "$1" in
		-t) shift; t=$1 ;; -m) shift; m=$1 ;; --tmpdir=*) tmp=${1#--tmpdir=} ;; -o) shift; out=$1 ;; *) in=$1 ;;
		esac; shift
	done
	[ "$in" = "/dev/stdin" ] && in=-
	# sambamba's -m is the total budget, samtools' is per thread
	g=${m%G}; per=$(( (g * 1024) / t )); [ $per -lt 64 ] && per=64
	exec "$ST" sort -@ "$t" -m "${per}M" -T "$tmp/srt" -o "$out" "$in" ;;
index) exec "$ST" index "$1" ;;
merge)
	t=1
	while [ $# -gt 0 ]; do case "$1" in -t) shift; t=$1; shift ;; *) break ;; esac; done
	out=$1; shift
	exec "$ST" merge -@ "$t" "$out" "$@" ;;
*) echo "sambamba shim: unsupported subcommand $cmd" >&2; exit 2 ;;
esac
