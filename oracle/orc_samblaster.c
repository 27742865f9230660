/*
 * oracle/orc_samblaster.c -- CPU ORACLE (test infrastructure): SAMBLASTER restatement.
 *
 * Restates upstream GregoryFaust/samblaster samblaster.cpp as invoked by the reference at
 * /root/reference/bin/speedseq:439 (`samblaster [--excludeDups] --addMateTags --maxSplitCount c
 * --minNonOverlap m --splitterFile F --discordantFile F`).  The source is absent from
 * /root/reference (empty submodule src/samblaster, .gitmodules:4-6); behaviour follows
 * SURVEY.md Appendix C.  PARITY UNPINNED.  The semantics fixed here (and mirrored bit-for-bit by
 * the HIP path) are:
 *   block      = consecutive lines with one QNAME; primaries = lines without 0x100/0x800,
 *                read1 = 0x40, read2 = 0x80.
 *   5' key     = (contig index, strand, unclipped 5' coordinate): forward POS - leading S/H,
 *                reverse POS + reflen(CIGAR) - 1 + trailing S/H.
 *   signature  = the two end keys ordered by (contig, coordinate, strand); pairs with one end
 *                unmapped use the mapped end only (separate key space); both unmapped: never dup.
 *   duplicate  = signature seen in an earlier block -> OR 0x400 into every line of the block.
 *   mate tags  = MC:Z:<mate primary CIGAR> and MQ:i:<mate primary MAPQ> appended when absent.
 *   discordant = both primaries mapped and 0x2 clear -> both primaries to the discordant stream.
 *   splitter   = per read, primary + 0x800/0x100 lines, 2..maxSplitCount of them, sorted by
 *                query start; adjacent pieces qualify when the non-overlapping query span is
 *                >= minNonOverlap and (same contig+strand: |ref gap - query gap| >= minIndelSize
 *                and unexplained unaligned bases <= maxUnmappedBases; otherwise unaligned
 *                bases between them <= maxUnmappedBases); QNAME gets _1/_2.
 *   --excludeDups keeps duplicate blocks out of both side streams.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "orc.h"

void orc_sbl_opt_init(orc_sbl_opt_t *o)
{
	o->exclude_dups = 0; o->add_mate_tags = 0; o->max_split_count = 2; o->min_non_overlap = 20;
	o->max_unmapped_bases = 50; o->min_indel_size = 50;
}

typedef struct {
	char *buf; size_t len;       /* the raw line without '\n' */
	char *f[12]; int nf;         /* first 11 fields (NUL-terminated copies live in fbuf) */
	char *fbuf; char *opt;       /* opt -> start of optional fields inside buf or NULL */
	char *extra;                 /* tags appended by --addMateTags */
	int flag, seq, pos, mapq, lclip, rclip, qalen, ralen, sqo, eqo, is_split;
} line_t;

typedef struct { uint64_t k[3]; } sig_t;
typedef struct { sig_t *a; uint8_t *used; size_t cap, n; } set_t;
static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static int set_insert(set_t *s, const sig_t *k)
{	/* returns 1 if newly inserted, 0 if already present */
	if ((s->n + 1) * 2 > s->cap) {
		size_t ncap = s->cap ? s->cap << 1 : 1 << 16; sig_t *na = malloc(ncap * sizeof(sig_t)); uint8_t *nu = calloc(ncap, 1);
		for (size_t i = 0; i < s->cap; ++i) if (s->used[i]) {
			size_t h = mix(s->a[i].k[0] ^ mix(s->a[i].k[1] ^ mix(s->a[i].k[2]))) & (ncap - 1);
			while (nu[h]) h = (h + 1) & (ncap - 1);
			nu[h] = 1; na[h] = s->a[i];
		}
		free(s->a); free(s->used); s->a = na; s->used = nu; s->cap = ncap;
	}
	size_t h = mix(k->k[0] ^ mix(k->k[1] ^ mix(k->k[2]))) & (s->cap - 1);
	while (s->used[h]) {
		if (memcmp(&s->a[h], k, sizeof(sig_t)) == 0) return 0;
		h = (h + 1) & (s->cap - 1);
	}
	s->used[h] = 1; s->a[h] = *k; ++s->n;
	return 1;
}

typedef struct { char **name; int n, m; } seqtab_t;
static int seq_lookup(const seqtab_t *t, const char *nm) { for (int i = 0; i < t->n; ++i) if (strcmp(t->name[i], nm) == 0) return i; return -1; }

static void parse_cigar(line_t *l)
{
	const char *c = l->f[5];
	l->lclip = l->rclip = l->qalen = l->ralen = 0;
	if (c[0] == '*') return;
	int first = 1;
	while (*c) {
		int n = (int)strtol(c, (char**)&c, 10); char op = *c++;
		switch (op) {
		case 'S': case 'H': if (first) l->lclip += n; else l->rclip += n; break;
		case 'M': case '=': case 'X': l->qalen += n; l->ralen += n; first = 0; break;
		case 'I': l->qalen += n; first = 0; break;
		case 'D': case 'N': l->ralen += n; first = 0; break;
		default: break;
		}
		if (op != 'S' && op != 'H') l->rclip = 0;
	}
	/* trailing clips: recount from the end */
	{
		const char *e = l->f[5]; int rc = 0; int n; char op;
		while (*e) { n = (int)strtol(e, (char**)&e, 10); op = *e++; if (op == 'S' || op == 'H') rc += n; else rc = 0; }
		l->rclip = (l->qalen + l->ralen) ? rc : 0;
	}
	if (l->flag & 0x10) { l->sqo = l->rclip; } else { l->sqo = l->lclip; }
	l->eqo = l->sqo + l->qalen - 1;
}

static int parse_line(line_t *l, const seqtab_t *st)
{
	free(l->fbuf); l->fbuf = malloc(l->len + 1); memcpy(l->fbuf, l->buf, l->len + 1);
	char *p = l->fbuf; l->nf = 0; l->opt = 0;
	while (l->nf < 11) {
		l->f[l->nf++] = p;
		char *t = strchr(p, '\t');
		if (!t) break;
		*t = 0; p = t + 1;
		if (l->nf == 11) l->opt = l->buf + (p - l->fbuf);
	}
	if (l->nf < 11) return -1;
	l->flag = atoi(l->f[1]); l->pos = atoi(l->f[3]); l->mapq = atoi(l->f[4]);
	l->seq = l->f[2][0] == '*' ? -1 : seq_lookup(st, l->f[2]);
	l->is_split = 0;
	parse_cigar(l);
	return 0;
}

static int has_tag(const line_t *l, const char *tag)
{
	const char *p = l->opt;
	while (p && *p) { if (strncmp(p, tag, 5) == 0) return 1; p = strchr(p, '\t'); if (p) ++p; }
	return 0;
}

static void key5(const line_t *l, uint64_t *seq, uint64_t *pos, uint64_t *strand)
{
	*seq = (uint64_t)l->seq; *strand = (l->flag & 0x10) ? 1 : 0;
	int64_t p = *strand ? (int64_t)l->pos + l->ralen - 1 + l->rclip : (int64_t)l->pos - l->lclip;
	*pos = (uint64_t)(p + (1LL << 31)); /* keep clipped-past-start coordinates non-negative */
}

static void write_line(FILE *fp, const line_t *l, int flag, const char *suffix, const char *extra)
{	/* re-emit with a (possibly) modified FLAG / QNAME suffix / appended tags */
	fputs(l->f[0], fp); if (suffix) fputs(suffix, fp);
	fprintf(fp, "\t%d", flag);
	for (int i = 2; i < 11; ++i) { fputc('\t', fp); fputs(l->f[i], fp); }
	if (l->opt) { fputc('\t', fp); fputs(l->opt, fp); }
	if (extra) fputs(extra, fp);
	fputc('\n', fp);
}

static int cmp_sqo(const void *a, const void *b) { const line_t *x = *(line_t* const*)a, *y = *(line_t* const*)b; return x->sqo < y->sqo ? -1 : x->sqo > y->sqo; }

static void mark_splitters(const orc_sbl_opt_t *o, line_t *blk, int n, int mask)
{
	line_t *arr[64]; int cnt = 0;
	for (int i = 0; i < n; ++i) if (blk[i].flag & mask) { if (cnt == 64) return; arr[cnt++] = &blk[i]; }
	if (cnt < 2 || cnt > o->max_split_count) return;
	for (int i = 0; i < cnt; ++i) if ((arr[i]->flag & 0x4) || arr[i]->seq < 0) return;
	/* stable order for equal SQO: insertion sort keeps block order */
	for (int i = 1; i < cnt; ++i) { line_t *t = arr[i]; int j = i; while (j > 0 && cmp_sqo(&arr[j-1], &t) > 0) { arr[j] = arr[j-1]; --j; } arr[j] = t; }
	line_t *left = arr[0];
	for (int i = 1; i < cnt; ++i) {
		line_t *right = arr[i];
		int lo = left->sqo > right->sqo ? left->sqo : right->sqo, hi = left->eqo < right->eqo ? left->eqo : right->eqo;
		int overlap = 1 + hi - lo; if (overlap < 0) overlap = 0;
		int alen1 = 1 + left->eqo - left->sqo, alen2 = 1 + right->eqo - right->sqo;
		int mno = (alen1 < alen2 ? alen1 : alen2) - overlap;
		int desert = right->sqo - left->eqo - 1, ok = 1;
		if (mno < o->min_non_overlap) ok = 0;
		else if (left->seq == right->seq && (left->flag & 0x10) == (right->flag & 0x10)) {
			int64_t ld, rd, ins;
			if (!(left->flag & 0x10)) { ld = (int64_t)left->pos - left->sqo; rd = (int64_t)right->pos - right->sqo; ins = rd - ld; }
			else { ld = (int64_t)left->pos + left->ralen - 1 + left->sqo; rd = (int64_t)right->pos + right->ralen - 1 + right->sqo; ins = ld - rd; }
			if (desert > 0 && desert - (ins > 0 ? ins : 0) > o->max_unmapped_bases) ok = 0;
			if ((ins < 0 ? -ins : ins) < o->min_indel_size) ok = 0;
		} else if (desert > o->max_unmapped_bases) ok = 0;
		if (ok) left->is_split = right->is_split = 1;
		left = right;
	}
}

static void process_block(const orc_sbl_opt_t *o, line_t *blk, int n, const seqtab_t *st, set_t *pairs, set_t *orphans,
                          FILE *out, FILE *spl, FILE *disc, uint64_t stats[4])
{
	line_t *r1 = 0, *r2 = 0; int dup = 0;
	for (int i = 0; i < n; ++i) parse_line(&blk[i], st);
	for (int i = 0; i < n; ++i) {
		if (blk[i].flag & (0x100 | 0x800)) continue;
		if ((blk[i].flag & 0x40) && !r1) r1 = &blk[i];
		else if ((blk[i].flag & 0x80) && !r2) r2 = &blk[i];
	}
	if (r1 && r2) {
		int m1 = !(r1->flag & 0x4) && r1->seq >= 0, m2 = !(r2->flag & 0x4) && r2->seq >= 0;
		++stats[0];
		if (m1 && m2) {
			uint64_t a[3], b[3]; sig_t k;
			key5(r1, &a[0], &a[1], &a[2]); key5(r2, &b[0], &b[1], &b[2]);
			int swap = a[0] > b[0] || (a[0] == b[0] && (a[1] > b[1] || (a[1] == b[1] && a[2] > b[2])));
			uint64_t *lo = swap ? b : a, *hi = swap ? a : b;
			k.k[0] = lo[0] << 32 | hi[0]; k.k[1] = lo[1] << 1 | lo[2]; k.k[2] = hi[1] << 1 | hi[2];
			dup = !set_insert(pairs, &k);
		} else if (m1 || m2) {
			uint64_t a[3]; sig_t k;
			key5(m1 ? r1 : r2, &a[0], &a[1], &a[2]);
			k.k[0] = a[0]; k.k[1] = a[1] << 1 | a[2]; k.k[2] = 0;
			dup = !set_insert(orphans, &k);
		}
		if (dup) ++stats[1];
	}
	for (int i = 0; i < n; ++i) {
		char extra[1024]; extra[0] = 0;
		if (dup) blk[i].flag |= 0x400;
		if (o->add_mate_tags && r1 && r2) {
			line_t *mate = (blk[i].flag & 0x40) ? r2 : (blk[i].flag & 0x80) ? r1 : 0;
			if (mate) {
				size_t l = 0;
				if (!has_tag(&blk[i], "MC:Z:")) l += snprintf(extra + l, sizeof(extra) - l, "\tMC:Z:%s", mate->f[5]);
				if (!has_tag(&blk[i], "MQ:i:")) l += snprintf(extra + l, sizeof(extra) - l, "\tMQ:i:%s", mate->f[4]);
			}
		}
		/* side streams see the line as written to stdout (flag + tags) */
		write_line(out, &blk[i], blk[i].flag, 0, extra);
		blk[i].is_split = 0;
		free(blk[i].extra); blk[i].extra = strdup(extra);
	}
	if (!(dup && o->exclude_dups) && r1 && r2) {
		if (disc && !(r1->flag & 0x4) && !(r2->flag & 0x4) && r1->seq >= 0 && r2->seq >= 0 && !(r1->flag & 0x2)) {
			write_line(disc, r1, r1->flag, 0, r1->extra); write_line(disc, r2, r2->flag, 0, r2->extra);
			++stats[2];
		}
		if (spl) {
			mark_splitters(o, blk, n, 0x40);
			mark_splitters(o, blk, n, 0x80);
			for (int i = 0; i < n; ++i) if (blk[i].is_split) { write_line(spl, &blk[i], blk[i].flag, (blk[i].flag & 0x40) ? "_1" : "_2", blk[i].extra); ++stats[3]; }
		}
	}
}

int orc_samblaster(const orc_sbl_opt_t *o, FILE *in, FILE *out, FILE *spl, FILE *disc, uint64_t stats[4])
{
	seqtab_t st = {0,0,0}; set_t pairs = {0,0,0,0}, orphans = {0,0,0,0};
	line_t *blk = 0; int nb = 0, mb = 0;
	char *line = 0; size_t cap = 0; ssize_t r; int in_header = 1;
	uint64_t st_local[4] = {0,0,0,0}; if (!stats) stats = st_local;
	const char *pg = "@PG\tID:SAMBLASTER\tVN:0.1.22-ssgpu\tCL:samblaster\n";
	while ((r = getline(&line, &cap, in)) > 0) {
		while (r > 0 && (line[r-1] == '\n' || line[r-1] == '\r')) line[--r] = 0;
		if (in_header && line[0] == '@') {
			if (strncmp(line, "@SQ", 3) == 0) {
				char *sn = strstr(line, "\tSN:");
				if (sn) {
					sn += 4; char *e = strchr(sn, '\t'); size_t l = e ? (size_t)(e - sn) : strlen(sn);
					if (st.n == st.m) { st.m = st.m ? st.m << 1 : 64; st.name = realloc(st.name, st.m * sizeof(char*)); }
					st.name[st.n++] = strndup(sn, l);
				}
			}
			fputs(line, out); fputc('\n', out);
			if (spl) { fputs(line, spl); fputc('\n', spl); }
			if (disc) { fputs(line, disc); fputc('\n', disc); }
			continue;
		}
		if (in_header) {
			in_header = 0;
			fputs(pg, out); if (spl) fputs(pg, spl); if (disc) fputs(pg, disc);
		}
		/* same QNAME as the current block? */
		size_t ql = strcspn(line, "\t");
		if (nb && !(strlen(blk[0].f[0]) == ql && strncmp(blk[0].f[0], line, ql) == 0)) {
			process_block(o, blk, nb, &st, &pairs, &orphans, out, spl, disc, stats);
			for (int i = 0; i < nb; ++i) { free(blk[i].buf); free(blk[i].fbuf); free(blk[i].extra); }
			nb = 0;
		}
		if (nb == mb) { mb = mb ? mb << 1 : 8; blk = realloc(blk, mb * sizeof(line_t)); }
		memset(&blk[nb], 0, sizeof(line_t));
		blk[nb].buf = strdup(line); blk[nb].len = r;
		if (nb == 0) { /* need f[0] for QNAME comparison */
			blk[0].fbuf = strndup(line, ql); blk[0].f[0] = blk[0].fbuf;
		}
		++nb;
	}
	if (in_header) { fputs(pg, out); if (spl) fputs(pg, spl); if (disc) fputs(pg, disc); }
	if (nb) {
		process_block(o, blk, nb, &st, &pairs, &orphans, out, spl, disc, stats);
		for (int i = 0; i < nb; ++i) { free(blk[i].buf); free(blk[i].fbuf); free(blk[i].extra); }
	}
	free(blk); free(line);
	for (int i = 0; i < st.n; ++i) free(st.name[i]);
	free(st.name); free(pairs.a); free(pairs.used); free(orphans.a); free(orphans.used);
	return 0;
}
