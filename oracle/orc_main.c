/*
 * oracle/orc_main.c -- CPU ORACLE command line (test infrastructure / cpu_baseline leg only).
 *
 *   orc_bwa index <ref.fa>
 *   orc_bwa mem [-t INT] [-p] [-R STR] [-I mean[,std[,max[,min]]]] [-C] <ref.fa> <in1.fq> [in2.fq]
 *   orc_bwa samblaster [--excludeDups] [--addMateTags] [--maxSplitCount N] [--minNonOverlap N]
 *                      [--splitterFile F] [--discordantFile F]
 *
 * Restates the upstream `bwa` / `samblaster` command lines the reference issues at
 * /root/reference/bin/speedseq:389,438-439 (main_mem in upstream fastmap.c: batches of
 * chunk_size*n_threads bases with an even read count; FASTQ parsing with the semantics of
 * /root/reference/src/samtools-1.3.1/htslib-1.3.1/htslib/kseq.h:189-229).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>
#include <unistd.h>
#include <zlib.h>
#include <sys/time.h>
#include "orc.h"

static double now(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + 1e-6 * tv.tv_usec; }

/* klib kseq_read (htslib/kseq.h:189-229) over gzgetc, restated call for call: the record grammar is whatever that routine accepts
 * -- '>' or '@' headers, name up to the first blank, the rest of the header line as comment, sequence lines until a line that
 * starts with '>', '+' or '@', quality lines until as many characters as bases (so a quality line may start with '@') -- and
 * upstream bseq.c's kseq2bseq1 + trim_readno on top. */
typedef struct { gzFile fp; int last_char; char *name, *comment, *seq, *qual; size_t nl, cl, sl, ql, nm, cm, sm, qm; } fq_t;
static void fq_putc(char **s, size_t *l, size_t *m, int c)
{
	if (*l + 2 > *m) { *m = *m ? *m << 1 : 256; *s = realloc(*s, *m); }
	(*s)[(*l)++] = (char)c; (*s)[*l] = 0;
}
/* ks_getuntil2: delimiter 0 = any blank (isspace), 2 = end of line; appends; returns the appended length or -1 at EOF with nothing read */
static int fq_getuntil(fq_t *f, int delim, char **s, size_t *l, size_t *m, int *dret)
{
	int c, got = 0; size_t l0 = *l;
	if (dret) *dret = 0;
	if (!*s) { *m = 256; *s = malloc(*m); (*s)[0] = 0; }
	for (;;) {
		c = gzgetc(f->fp);
		if (c == -1) break;
		got = 1;
		if (delim == 2 ? c == '\n' : isspace(c)) { if (dret) *dret = c; break; }
		fq_putc(s, l, m, c);
	}
	if (!got) return -1;
	if (delim == 2 && *l > 1 && (*s)[*l - 1] == '\r') (*s)[--*l] = 0;
	return (int)(*l - l0);
}
static uint8_t nt4(int c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }

static int fq_kseq_read(fq_t *f)
{
	int c;
	if (f->last_char == 0) {
		while ((c = gzgetc(f->fp)) != -1 && c != '>' && c != '@') {}
		if (c == -1) return -1;
		f->last_char = c;
	}
	f->nl = f->cl = f->sl = f->ql = 0;
	if (f->name) f->name[0] = 0;
	if (f->comment) f->comment[0] = 0;
	if (fq_getuntil(f, 0, &f->name, &f->nl, &f->nm, &c) < 0) return -1;
	if (c != '\n') fq_getuntil(f, 2, &f->comment, &f->cl, &f->cm, 0);
	while ((c = gzgetc(f->fp)) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		fq_putc(&f->seq, &f->sl, &f->sm, c);
		fq_getuntil(f, 2, &f->seq, &f->sl, &f->sm, 0);
	}
	if (c == '>' || c == '@') f->last_char = c;
	if (c != '+') return (int)f->sl;
	while ((c = gzgetc(f->fp)) != -1 && c != '\n') {}
	if (c == -1) return -2;
	while (fq_getuntil(f, 2, &f->qual, &f->ql, &f->qm, 0) >= 0 && f->ql < f->sl) {}
	f->last_char = 0;
	if (f->sl != f->ql) return -2;
	return (int)f->sl;
}

static int fq_read1(fq_t *f, orc_read_t *r, int keep_comment)
{	/* kseq_read, then upstream kseq2bseq1 (comment kept only with -C and when not empty) and trim_readno */
	const int l = fq_kseq_read(f);
	if (l < 0) return l;
	r->name = strdup(f->name ? f->name : "");
	r->comment = (keep_comment && f->cl) ? strdup(f->comment) : 0;
	{
		size_t nl = strlen(r->name);
		if (nl > 2 && r->name[nl-2] == '/' && isdigit((unsigned char)r->name[nl-1])) r->name[nl-2] = 0;
	}
	r->l_seq = l; r->seq = malloc((size_t)l + 1);
	for (int i = 0; i < l; ++i) r->seq[i] = nt4(f->seq[i]);
	r->qual = f->ql ? strdup(f->qual) : 0;
	r->sam = 0;
	return 0;
}

static char *unescape_rg(const char *s)
{	/* upstream bwa_set_rg/bwa_escape: "\t" -> TAB */
	char *o = malloc(strlen(s) + 1), *q = o;
	for (const char *p = s; *p; ++p) {
		if (*p == '\\' && p[1] == 't') { *q++ = '\t'; ++p; }
		else if (*p == '\\' && p[1] == 'n') { *q++ = '\n'; ++p; }
		else if (*p == '\\' && p[1] == '\\') { *q++ = '\\'; ++p; }
		else *q++ = *p;
	}
	*q = 0;
	return o;
}

static int main_index(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: orc_bwa index <ref.fa>\n"); return 1; }
	orc_idx_t *idx = orc_idx_build_fasta(argv[argc-1]);
	if (!idx) { fprintf(stderr, "[orc_bwa] cannot read %s\n", argv[argc-1]); return 1; }
	int rc = orc_idx_save(idx, argv[argc-1]);
	orc_idx_destroy(idx);
	return rc ? 1 : 0;
}

static int main_mem(int argc, char **argv)
{
	orc_opt_t opt; orc_opt_init(&opt);
	int interleaved = 0, keep_comment = 0, fixed_chunk = 0, i; char *rg = 0, rg_id[256] = ""; orc_pestat_t pes0[4], *pes = 0;
	int ai, c; char *p;
	orc_opt_t opt0; memset(&opt0, 0, sizeof(opt0));   /* upstream main_mem: which scoring fields the command line set itself */
	optind = 1;
	/* upstream main_mem's option loop (fastmap.c, 0.7.12): same letters, same two-number forms, same order-independent scaling by -A below */
	while ((c = getopt(argc, argv, "pMCSPYk:c:v:s:r:t:R:A:B:O:E:U:w:L:d:T:Q:D:m:I:N:W:G:h:y:K:X:")) >= 0) {
		if (c == 'p') interleaved = 1;
		else if (c == 'C') keep_comment = 1;
		else if (c == 'M') opt.flag |= ORC_F_NO_MULTI;
		else if (c == 'Y') opt.flag |= ORC_F_SOFTCLIP;
		else if (c == 'S') opt.flag |= ORC_F_NO_RESCUE;
		else if (c == 'P') opt.flag |= ORC_F_NOPAIRING;
		else if (c == 'k') opt.min_seed_len = atoi(optarg), opt0.min_seed_len = 1;
		else if (c == 'w') opt.w = atoi(optarg), opt0.w = 1;
		else if (c == 'A') opt.a = atoi(optarg), opt0.a = 1;
		else if (c == 'B') opt.b = atoi(optarg), opt0.b = 1;
		else if (c == 'T') opt.T = atoi(optarg), opt0.T = 1;
		else if (c == 'U') opt.pen_unpaired = atoi(optarg), opt0.pen_unpaired = 1;
		else if (c == 't') opt.n_threads = atoi(optarg), opt.n_threads = opt.n_threads > 1 ? opt.n_threads : 1;
		else if (c == 'c') opt.max_occ = atoi(optarg), opt0.max_occ = 1;
		else if (c == 'd') opt.zdrop = atoi(optarg), opt0.zdrop = 1;
		else if (c == 'v') ;   /* verbosity: nothing on stdout depends on it */
		else if (c == 'r') opt.split_factor = atof(optarg), opt0.split_factor = 1.;
		else if (c == 'D') opt.drop_ratio = atof(optarg), opt0.drop_ratio = 1.;
		else if (c == 'm') opt.max_matesw = atoi(optarg), opt0.max_matesw = 1;
		else if (c == 's') opt.split_width = atoi(optarg), opt0.split_width = 1;
		else if (c == 'G') opt.max_chain_gap = atoi(optarg), opt0.max_chain_gap = 1;
		else if (c == 'N') opt.max_chain_extend = atoi(optarg), opt0.max_chain_extend = 1;
		else if (c == 'W') opt.min_chain_weight = atoi(optarg), opt0.min_chain_weight = 1;
		else if (c == 'y') opt.max_mem_intv = atol(optarg), opt0.max_mem_intv = 1;
		else if (c == 'K') fixed_chunk = atoi(optarg);
		else if (c == 'X') opt.mask_level = atof(optarg);
		else if (c == 'h') {
			opt.max_XA_hits = opt.max_XA_hits_alt = strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt.max_XA_hits_alt = strtol(p + 1, &p, 10);
		} else if (c == 'Q') {
			opt.mapQ_coef_len = atoi(optarg);
			opt.mapQ_coef_fac = opt.mapQ_coef_len > 0 ? log(opt.mapQ_coef_len) : 0;
		} else if (c == 'O') {
			opt0.o_del = opt0.o_ins = 1;
			opt.o_del = opt.o_ins = strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt.o_ins = strtol(p + 1, &p, 10);
		} else if (c == 'E') {
			opt0.e_del = opt0.e_ins = 1;
			opt.e_del = opt.e_ins = strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt.e_ins = strtol(p + 1, &p, 10);
		} else if (c == 'L') {
			opt0.pen_clip5 = opt0.pen_clip3 = 1;
			opt.pen_clip5 = opt.pen_clip3 = strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt.pen_clip3 = strtol(p + 1, &p, 10);
		} else if (c == 'R') rg = unescape_rg(optarg);
		else if (c == 'I') { /* upstream main_mem -I: FR only */
			pes = pes0; memset(pes0, 0, sizeof(pes0));
			pes0[0].failed = pes0[2].failed = pes0[3].failed = 1;
			pes0[1].avg = strtod(optarg, &p);
			pes0[1].std = pes0[1].avg * .1;
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0[1].std = strtod(p + 1, &p);
			pes0[1].high = (int)(pes0[1].avg + 4. * pes0[1].std + .499);
			pes0[1].low  = (int)(pes0[1].avg - 4. * pes0[1].std + .499);
			if (pes0[1].low < 1) pes0[1].low = 1;
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0[1].high = (int)(strtod(p + 1, &p) + .499);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0[1].low  = (int)(strtod(p + 1, &p) + .499);
		} else { fprintf(stderr, "[orc_bwa] option outside the oracle's scope (upstream's -a -e -F -V -j -x -H are not restated)\n"); return 1; }
	}
	ai = optind;
	if (opt0.a) { /* upstream update_a: a changed match score scales what the command line left alone */
		if (!opt0.b) opt.b *= opt.a;
		if (!opt0.T) opt.T *= opt.a;
		if (!opt0.o_del) opt.o_del *= opt.a;
		if (!opt0.e_del) opt.e_del *= opt.a;
		if (!opt0.o_ins) opt.o_ins *= opt.a;
		if (!opt0.e_ins) opt.e_ins *= opt.a;
		if (!opt0.zdrop) opt.zdrop *= opt.a;
		if (!opt0.pen_clip5) opt.pen_clip5 *= opt.a;
		if (!opt0.pen_clip3) opt.pen_clip3 *= opt.a;
		if (!opt0.pen_unpaired) opt.pen_unpaired *= opt.a;
	}
	orc_fill_scmat(&opt);
	if (argc - ai < 2) { fprintf(stderr, "usage: orc_bwa mem [opts] <ref> <fq1> [fq2]\n"); return 1; }
	if (rg) {
		if (strncmp(rg, "@RG", 3) != 0) { fprintf(stderr, "[E::bwa_set_rg] the read group line is not started with @RG\n"); return 1; }
		char *p = strstr(rg, "\tID:");
		if (!p) { fprintf(stderr, "[E::bwa_set_rg] no ID at the read group line\n"); return 1; }
		p += 4; for (i = 0; p[i] && p[i] != '\t' && p[i] != '\n' && i < 255; ++i) rg_id[i] = p[i];
		rg_id[i] = 0;
	}
	double t0 = now();
	orc_idx_t *idx = orc_idx_load(argv[ai]);
	if (!idx) { fprintf(stderr, "[orc_bwa] fail to load index %s\n", argv[ai]); return 1; }
	fq_t f1, f2; memset(&f1, 0, sizeof(f1)); memset(&f2, 0, sizeof(f2));
	f1.fp = gzopen(argv[ai+1], "r");
	if (!f1.fp) { fprintf(stderr, "[orc_bwa] fail to open %s\n", argv[ai+1]); return 1; }
	if (argc - ai >= 3) { f2.fp = gzopen(argv[ai+2], "r"); if (!f2.fp) { fprintf(stderr, "[orc_bwa] fail to open %s\n", argv[ai+2]); return 1; } }
	if (interleaved && f2.fp) { fprintf(stderr, "[W::main_mem] when '-p' is in use, the second query file is ignored.\n"); gzclose(f2.fp); f2.fp = 0; }
	const int se = !interleaved && !f2.fp;   /* upstream main_mem: MEM_F_PE is set by -p or by a second file */
	{
		char cl[4096]; size_t l = 0; cl[0] = 0;
		for (i = 0; i < argc && l < sizeof(cl) - 1; ++i) l += snprintf(cl + l, sizeof(cl) - l, "%s%s", i ? " " : "bwa ", argv[i]);
		char *h = orc_sam_header(idx, rg, cl); fputs(h, stdout); free(h);
	}
	{ const char *e = getenv("ORC_CHUNK_BASES"); if (e && atoi(e) > 0) opt.chunk_size = atoi(e); }   /* tests: many batches from a small input */
	int64_t n_processed = 0, chunk = fixed_chunk > 0 ? fixed_chunk : (int64_t)opt.chunk_size * opt.n_threads;   /* -K: batches that do not depend on -t */
	for (;;) {
		orc_read_t *s = 0; int n = 0, m = 0; int64_t size = 0; int rc = 0;
		while (1) { /* upstream bseq_read */
			if (n + 2 > m) { m = m ? m << 1 : 1024; s = realloc(s, m * sizeof(orc_read_t)); }
			rc = fq_read1(&f1, &s[n], keep_comment); if (rc < 0) break;
			size += s[n++].l_seq;
			if (f2.fp) {
				rc = fq_read1(&f2, &s[n], keep_comment);
				if (rc == -1) { /* upstream bseq_read: the complete pairs read so far are kept */
					fprintf(stderr, "[W::bseq_read] the 2nd file has fewer sequences.\n");
					--n; free(s[n].name); free(s[n].comment); free(s[n].seq); free(s[n].qual); break;
				}
				if (rc < 0) { rc = -2; break; }
				size += s[n++].l_seq;
			}
			if (size >= chunk && (n & 1) == 0) break;
		}
		if (rc == -2) { fprintf(stderr, "[orc_bwa] truncated / malformed FASTQ\n"); return 1; }
		if (se) { /* upstream mem_process_seqs without MEM_F_PE */
			if (n == 0) { free(s); break; }
			orc_mem_process_reads(&opt, idx, n_processed, n, s, rg_id, opt.n_threads);
			for (i = 0; i < n; ++i) {
				fputs(s[i].sam, stdout);
				free(s[i].sam); free(s[i].name); free(s[i].comment); free(s[i].seq); free(s[i].qual);
			}
			n_processed += n;
			free(s);
			if (rc < 0) break;
			continue;
		}
		if (n & 1) { /* upstream main_mem, PE mode: the odd read at the end of the input is dropped */
			fprintf(stderr, "[W::main_mem] odd number of reads in the PE mode; last read dropped\n");
			--n; free(s[n].name); free(s[n].comment); free(s[n].seq); free(s[n].qual);
		}
		if (n == 0) { free(s); break; }
		for (i = 0; i < n; i += 2)
			if (strcmp(s[i].name, s[i+1].name) != 0) { fprintf(stderr, "[mem_sam_pe] paired reads have different names: \"%s\", \"%s\"\n", s[i].name, s[i+1].name); return 1; }
		orc_pestat_t pes_out[4];
		orc_mem_process_pairs(&opt, idx, n_processed, n, s, pes, rg_id, pes_out, opt.n_threads);
		fprintf(stderr, "[orc_bwa] processed %d reads; FR insert: failed=%d low=%d high=%d avg=%.2f std=%.2f\n", n, pes_out[1].failed, pes_out[1].low, pes_out[1].high, pes_out[1].avg, pes_out[1].std);
		for (i = 0; i < n; ++i) {
			fputs(s[i].sam, stdout);
			free(s[i].sam); free(s[i].name); free(s[i].comment); free(s[i].seq); free(s[i].qual);
		}
		n_processed += n;
		free(s);
		if (rc < 0) break;
	}
	fprintf(stderr, "[orc_bwa] %lld reads in %.3f s; cells=%llu extend=%llu lf=%llu sa=%llu\n", (long long)n_processed, now() - t0,
	        (unsigned long long)orc_tot_cells, (unsigned long long)orc_tot_extend, (unsigned long long)orc_tot_lf, (unsigned long long)orc_tot_sa);
	gzclose(f1.fp); if (f2.fp) gzclose(f2.fp);
	orc_idx_destroy(idx);
	return 0;
}

static int main_samblaster(int argc, char **argv)
{
	orc_sbl_opt_t o; orc_sbl_opt_init(&o);
	const char *spl = 0, *disc = 0;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--excludeDups")) o.exclude_dups = 1;
		else if (!strcmp(argv[i], "--addMateTags")) o.add_mate_tags = 1;
		else if (!strcmp(argv[i], "--maxSplitCount") && i + 1 < argc) o.max_split_count = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--minNonOverlap") && i + 1 < argc) o.min_non_overlap = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--splitterFile") && i + 1 < argc) spl = argv[++i];
		else if (!strcmp(argv[i], "--discordantFile") && i + 1 < argc) disc = argv[++i];
		else { fprintf(stderr, "[orc_samblaster] unsupported option %s\n", argv[i]); return 1; }
	}
	FILE *fs = spl ? fopen(spl, "w") : 0, *fd = disc ? fopen(disc, "w") : 0;
	uint64_t st[4] = {0,0,0,0};
	int rc = orc_samblaster(&o, stdin, stdout, fs, fd, st);
	if (fs) fclose(fs);
	if (fd) fclose(fd);
	fprintf(stderr, "[orc_samblaster] pairs=%llu dups=%llu discordant_pairs=%llu splitter_lines=%llu\n",
	        (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3]);
	return rc;
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: orc_bwa <index|mem|samblaster> ...\n"); return 1; }
	if (!strcmp(argv[1], "index")) return main_index(argc - 1, argv + 1);
	if (!strcmp(argv[1], "mem")) return main_mem(argc - 1, argv + 1);
	if (!strcmp(argv[1], "samblaster")) return main_samblaster(argc - 1, argv + 1);
	fprintf(stderr, "unknown command %s\n", argv[1]);
	return 1;
}
