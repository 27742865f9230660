/*
 * oracle/orc_pair.c -- CPU ORACLE (test infrastructure): BWA-MEM paired-end stage and SAM output.
 *
 * Restates upstream lh3/bwa bwamem_pair.c (mem_infer_dir, cal_sub, mem_pestat, mem_matesw,
 * mem_pair, mem_sam_pe), bwamem.c (mem_mark_primary_se, mem_approx_mapq_se, mem_reg2aln,
 * mem_aln2sam, mem_reg2sam, mem_process_seqs), bwamem_extra.c (mem_gen_alt -> XA tag) and
 * bwa.c (bwa_gen_cigar2, bwa_print_sam_hdr).  Upstream sources are absent from
 * /root/reference; behaviour follows SURVEY.md Appendix B (0.7.12-era: no MC tag from bwa,
 * samblaster --addMateTags supplies MC/MQ as in bin/speedseq:439).  PARITY UNPINNED.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <assert.h>
#include <pthread.h>
#include <malloc.h>
#include <time.h>
#include "orc.h"

#define PUSH(v, T, x) do { if ((v).n == (v).m) { (v).m = (v).m ? (v).m << 1 : 4; (v).a = realloc((v).a, sizeof(T) * (v).m); } (v).a[(v).n++] = (x); } while (0)

/* ---------- tiny string builder (kstring semantics) ---------- */
typedef struct { size_t l, m; char *s; } str_t;
static void s_need(str_t *s, size_t add) { if (s->l + add + 1 > s->m) { s->m = (s->l + add + 1) * 2; s->s = realloc(s->s, s->m); } }
static void s_putc(str_t *s, int c) { s_need(s, 1); s->s[s->l++] = c; s->s[s->l] = 0; }
static void s_putsn(str_t *s, const char *p, size_t n) { s_need(s, n); memcpy(s->s + s->l, p, n); s->l += n; s->s[s->l] = 0; }
static void s_puts(str_t *s, const char *p) { s_putsn(s, p, strlen(p)); }
static void s_putl(str_t *s, long long v) { char b[32]; int n = snprintf(b, 32, "%lld", v); s_putsn(s, b, n); }

/* ---------- insert-size statistics ---------- */
#define MIN_RATIO     0.8
#define MIN_DIR_CNT   10
#define MIN_DIR_RATIO 0.05
#define OUTLIER_BOUND 2.0
#define MAPPING_BOUND 3.0
#define MAX_STDDEV    4.0

static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{	/* upstream mem_infer_dir */
	int64_t p2;
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

static int cal_sub(const orc_opt_t *opt, const orc_alnreg_v *r)
{
	size_t j;
	for (j = 1; j < r->n; ++j) {
		int b_max = r->a[j].qb > r->a[0].qb ? r->a[j].qb : r->a[0].qb;
		int e_min = r->a[j].qe < r->a[0].qe ? r->a[j].qe : r->a[0].qe;
		if (e_min > b_max) {
			int min_l = r->a[j].qe - r->a[j].qb < r->a[0].qe - r->a[0].qb ? r->a[j].qe - r->a[j].qb : r->a[0].qe - r->a[0].qb;
			if (e_min - b_max >= min_l * opt->mask_level) break;
		}
	}
	return j < r->n ? r->a[j].score : opt->min_seed_len * opt->a;
}

void orc_mem_pestat(const orc_opt_t *opt, int64_t l_pac, int n, const orc_alnreg_v *regs, orc_pestat_t pes[4])
{	/* upstream mem_pestat */
	int i, d, max;
	struct { size_t n, m; uint64_t *a; } isize[4];
	memset(pes, 0, 4 * sizeof(orc_pestat_t));
	memset(isize, 0, sizeof(isize));
	for (i = 0; i < n >> 1; ++i) {
		int dir; int64_t is;
		const orc_alnreg_v *r[2] = { &regs[i<<1|0], &regs[i<<1|1] };
		if (r[0]->n == 0 || r[1]->n == 0) continue;
		if (cal_sub(opt, r[0]) > MIN_RATIO * r[0]->a[0].score) continue;
		if (cal_sub(opt, r[1]) > MIN_RATIO * r[1]->a[0].score) continue;
		if (r[0]->a[0].rid != r[1]->a[0].rid) continue;
		dir = infer_dir(l_pac, r[0]->a[0].rb, r[1]->a[0].rb, &is);
		if (is && is <= opt->max_ins) PUSH(isize[dir], uint64_t, (uint64_t)is);
	}
	for (d = 0; d < 4; ++d) {
		orc_pestat_t *r = &pes[d];
		int p25, p50, p75, x; size_t k;
		if (isize[d].n < MIN_DIR_CNT) { r->failed = 1; continue; }
		orc_introsort_u64(isize[d].n, isize[d].a);
		p25 = (int)isize[d].a[(int)(.25 * isize[d].n + .499)];
		p50 = (int)isize[d].a[(int)(.50 * isize[d].n + .499)];
		p75 = (int)isize[d].a[(int)(.75 * isize[d].n + .499)];
		(void)p50;
		r->low  = (int)(p25 - OUTLIER_BOUND * (p75 - p25) + .499);
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + OUTLIER_BOUND * (p75 - p25) + .499);
		for (k = 0, x = 0, r->avg = 0; k < isize[d].n; ++k)
			if (isize[d].a[k] >= (uint64_t)r->low && isize[d].a[k] <= (uint64_t)r->high) r->avg += isize[d].a[k], ++x;
		r->avg /= x;
		for (k = 0, r->std = 0; k < isize[d].n; ++k)
			if (isize[d].a[k] >= (uint64_t)r->low && isize[d].a[k] <= (uint64_t)r->high)
				r->std += (isize[d].a[k] - r->avg) * (isize[d].a[k] - r->avg);
		r->std = sqrt(r->std / x);
		r->low  = (int)(p25 - MAPPING_BOUND * (p75 - p25) + .499);
		r->high = (int)(p75 + MAPPING_BOUND * (p75 - p25) + .499);
		if (r->low  > r->avg - MAX_STDDEV * r->std) r->low  = (int)(r->avg - MAX_STDDEV * r->std + .499);
		if (r->high < r->avg + MAX_STDDEV * r->std) r->high = (int)(r->avg + MAX_STDDEV * r->std + .499);
		if (r->low < 1) r->low = 1;
	}
	for (d = 0, max = 0; d < 4; ++d) max = max > (int)isize[d].n ? max : (int)isize[d].n;
	for (d = 0; d < 4; ++d)
		if (pes[d].failed == 0 && isize[d].n < max * MIN_DIR_RATIO) pes[d].failed = 1;
	for (d = 0; d < 4; ++d) free(isize[d].a);
}

/* ---------- mate rescue ---------- */
static uint8_t *fetch_seq2(const orc_idx_t *idx, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{	/* upstream bns_fetch_seq (same as in orc_mem.c) */
	const orc_bns_t *bns = idx->bns;
	int64_t far_beg, far_end; int is_rev;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	*rid = orc_bns_pos2rid(bns, orc_bns_depos(bns, mid, &is_rev));
	far_beg = bns->anns[*rid].offset;
	far_end = far_beg + bns->anns[*rid].len;
	if (is_rev) { int64_t t = far_beg; far_beg = (bns->l_pac << 1) - far_end; far_end = (bns->l_pac << 1) - t; }
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	if (*end <= *beg) return calloc(1, 1);
	uint8_t *seq = malloc(*end - *beg + 1);
	for (int64_t k = *beg; k < *end; ++k) seq[k - *beg] = orc_ref_base(idx->pac, bns->l_pac, k);
	return seq;
}

int orc_mem_matesw(const orc_opt_t *opt, const orc_idx_t *idx, const orc_pestat_t pes[4], const orc_alnreg_t *a, int l_ms, const uint8_t *ms, orc_alnreg_v *ma)
{	/* upstream mem_matesw */
	int64_t l_pac = idx->bns->l_pac;
	int i, r, skip[4], n = 0, rid = -1;
	for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (i = 0; i < (int)ma->n; ++i) {
		int64_t dist;
		r = infer_dir(l_pac, a->rb, ma->a[i].rb, &dist);
		if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (r = 0; r < 4; ++r) {
		int is_rev, is_larger;
		uint8_t *seq, *rev = 0, *ref = 0;
		int64_t rb, re;
		if (skip[r]) continue;
		is_rev = (r >> 1 != (r & 1));
		is_larger = !(r >> 1);
		if (is_rev) {
			rev = malloc(l_ms);
			for (i = 0; i < l_ms; ++i) rev[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
			seq = rev;
		} else seq = (uint8_t*)ms;
		if (!is_rev) {
			rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
			re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
			re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) ref = fetch_seq2(idx, &rb, (rb + re) >> 1, &re, &rid);
		if (a->rid == rid && re - rb >= opt->min_seed_len) {
			orc_kswr_t aln; orc_alnreg_t b; int tmp;
			int xtra = ORC_KSW_XSUBO | ORC_KSW_XSTART | (l_ms * opt->a < 250 ? ORC_KSW_XBYTE : 0) | (opt->min_seed_len * opt->a);
			uint8_t *qcopy = malloc(l_ms); memcpy(qcopy, seq, l_ms);
			aln = orc_ksw_align2(l_ms, qcopy, (int)(re - rb), ref, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, xtra);
			free(qcopy);
			memset(&b, 0, sizeof(b));
			if (aln.score >= opt->min_seed_len && aln.qb >= 0) {
				b.rid = a->rid; b.is_alt = a->is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score;
				b.csub = aln.score2;
				b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				PUSH(*ma, orc_alnreg_t, b);
				for (i = 0; i < (int)ma->n - 1; ++i) if (ma->a[i].score < b.score) break;
				tmp = i;
				for (i = (int)ma->n - 1; i > tmp; --i) ma->a[i] = ma->a[i-1];
				ma->a[i] = b;
			}
			++n;
		}
		if (n) ma->n = orc_mem_sort_dedup_patch(opt, 0, 0, (int)ma->n, ma->a);
		free(rev); free(ref);
	}
	return n;
}

/* ---------- primary marking / MAPQ ---------- */
static inline uint64_t hash_64(uint64_t key)
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}
static int hlt(const void *a_, const void *b_)
{
	const orc_alnreg_t *a = a_, *b = b_;
	return a->score > b->score || (a->score == b->score && (a->is_alt < b->is_alt || (a->is_alt == b->is_alt && a->hash < b->hash)));
}

int orc_mem_mark_primary_se(const orc_opt_t *opt, int n, orc_alnreg_t *a, int64_t id)
{	/* upstream mem_mark_primary_se (+ _core), ALT-free */
	int i, k, tmp, n_pri;
	struct { size_t n, m; int *a; } z = {0,0,0};
	if (n == 0) return 0;
	for (i = n_pri = 0; i < n; ++i) {
		a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash_64(id + i);
		if (!a[i].is_alt) ++n_pri;
	}
	orc_introsort(a, n, sizeof(orc_alnreg_t), hlt);
	tmp = opt->a + opt->b;
	tmp = opt->o_del + opt->e_del > tmp ? opt->o_del + opt->e_del : tmp;
	tmp = opt->o_ins + opt->e_ins > tmp ? opt->o_ins + opt->e_ins : tmp;
	PUSH(z, int, 0);
	for (i = 1; i < n; ++i) {
		for (k = 0; k < (int)z.n; ++k) {
			int j = z.a[k];
			int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
			int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
			if (e_min > b_max) {
				int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
				if (e_min - b_max >= min_l * opt->mask_level) {
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == (int)z.n) PUSH(z, int, i);
		else a[i].secondary = z.a[k];
	}
	for (i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
	free(z.a);
	return n_pri;
}

int orc_mem_approx_mapq_se(const orc_opt_t *opt, const orc_alnreg_t *a)
{	/* upstream mem_approx_mapq_se */
	int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
	double identity;
	sub = a->csub > sub ? a->csub : sub;
	if (sub >= a->score) return 0;
	l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
	identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
	if (a->score == 0) mapq = 0;
	else if (opt->mapQ_coef_len > 0) {
		double tmp;
		tmp = l < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
	} else {
		mapq = (int)(30.0 * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a->frac_rep) + .499);
	return mapq;
}

/* ---------- pairing ---------- */
typedef struct { uint64_t x, y; } pair64_t;
static int p128_lt(const void *a_, const void *b_) { const pair64_t *a = a_, *b = b_; return a->x < b->x || (a->x == b->x && a->y < b->y); }

int orc_mem_pair(const orc_opt_t *opt, const orc_idx_t *idx, const orc_pestat_t pes[4], const orc_alnreg_v a[2], int id, int *sub, int *n_sub, int z[2], int n_pri[2])
{	/* upstream mem_pair */
	struct { size_t n, m; pair64_t *a; } v = {0,0,0}, u = {0,0,0};
	int r, i, k, y[4], ret;
	int64_t l_pac = idx->bns->l_pac;
	for (r = 0; r < 2; ++r)
		for (i = 0; i < n_pri[r]; ++i) {
			pair64_t key;
			const orc_alnreg_t *e = &a[r].a[i];
			key.x = e->rb < l_pac ? e->rb : (l_pac << 1) - 1 - e->rb;
			key.x = (uint64_t)e->rid << 32 | (key.x - idx->bns->anns[e->rid].offset);
			key.y = (uint64_t)e->score << 32 | i << 2 | (e->rb >= l_pac) << 1 | r;
			PUSH(v, pair64_t, key);
		}
	orc_introsort(v.a, v.n, sizeof(pair64_t), p128_lt);
	y[0] = y[1] = y[2] = y[3] = -1;
	for (i = 0; i < (int)v.n; ++i) {
		for (r = 0; r < 2; ++r) {
			int dir = r << 1 | (v.a[i].y >> 1 & 1), which;
			if (pes[dir].failed) continue;
			which = r << 1 | ((v.a[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (k = y[which]; k >= 0; --k) {
				int64_t dist; int q; double ns; pair64_t p;
				if ((int)(v.a[k].y & 3) != which) continue;
				dist = (int64_t)v.a[i].x - v.a[k].x;
				if (dist > pes[dir].high) break;
				if (dist < pes[dir].low) continue;
				ns = (dist - pes[dir].avg) / pes[dir].std;
				q = (int)((v.a[i].y >> 32) + (v.a[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt->a + .499);
				if (q < 0) q = 0;
				p.y = (uint64_t)k << 32 | i;
				p.x = (uint64_t)q << 32 | (hash_64(p.y ^ (uint64_t)(id << 8)) & 0xffffffffU);
				PUSH(u, pair64_t, p);
			}
		}
		y[v.a[i].y & 3] = i;
	}
	if (u.n) {
		int tmp = opt->a + opt->b;
		tmp = tmp > opt->o_del + opt->e_del ? tmp : opt->o_del + opt->e_del;
		tmp = tmp > opt->o_ins + opt->e_ins ? tmp : opt->o_ins + opt->e_ins;
		orc_introsort(u.a, u.n, sizeof(pair64_t), p128_lt);
		i = u.a[u.n-1].y >> 32; k = u.a[u.n-1].y << 32 >> 32;
		z[v.a[i].y & 1] = v.a[i].y << 32 >> 34;
		z[v.a[k].y & 1] = v.a[k].y << 32 >> 34;
		ret = u.a[u.n-1].x >> 32;
		*sub = u.n > 1 ? u.a[u.n-2].x >> 32 : 0;
		for (i = (long)u.n - 2, *n_sub = 0; i >= 0; --i)
			if (*sub - (int)(u.a[i].x >> 32) <= tmp) ++*n_sub;
	} else ret = 0, *sub = 0, *n_sub = 0;
	free(u.a); free(v.a);
	return ret;
}

/* ---------- CIGAR / NM / MD ---------- */
uint32_t *orc_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac,
                         int l_query, uint8_t *query, int64_t rb, int64_t re, int *score, int *n_cigar, int *NM)
{	/* upstream bwa_gen_cigar2 */
	uint32_t *cigar = 0; uint8_t tmp, *rseq; int i; int64_t rlen;
	if (n_cigar) *n_cigar = 0;
	if (NM) *NM = -1;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
	if (rb < 0 || re > l_pac << 1) return 0; /* upstream: bns_get_seq truncates and the length check fails */
	rlen = re - rb;
	rseq = malloc(rlen + 1);
	for (int64_t k = rb; k < re; ++k) rseq[k - rb] = orc_ref_base(pac, l_pac, k);
	if (rb >= l_pac) { /* put both on the forward strand so that indels are left-aligned */
		for (i = 0; i < l_query >> 1; ++i) tmp = query[i], query[i] = query[l_query-1-i], query[l_query-1-i] = tmp;
		for (i = 0; i < rlen >> 1; ++i) tmp = rseq[i], rseq[i] = rseq[rlen-1-i], rseq[rlen-1-i] = tmp;
	}
	if (l_query == re - rb && w_ == 0) { /* gap-free */
		if (n_cigar) { cigar = malloc(4); cigar[0] = l_query << 4 | 0; *n_cigar = 1; }
		for (i = 0, *score = 0; i < l_query; ++i) *score += mat[rseq[i] * 5 + query[i]];
	} else {
		int w, max_gap, max_ins, max_del, min_w;
		max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
		max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
		max_gap = max_ins > max_del ? max_ins : max_del;
		max_gap = max_gap > 1 ? max_gap : 1;
		w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		min_w = abs((int)rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		*score = orc_ksw_global2(l_query, query, (int)rlen, rseq, 5, mat, o_del, e_del, o_ins, e_ins, w, n_cigar, n_cigar ? &cigar : 0);
	}
	if (NM && n_cigar) { /* NM and MD (MD stored after the cigar ops) */
		int k, x, y, u, n_mm = 0, n_gap = 0;
		str_t str = { (size_t)*n_cigar * 4, (size_t)*n_cigar * 4, (char*)cigar };
		const char *int2base = rb < l_pac ? "ACGTN" : "TGCAN";
		for (k = 0, x = y = u = 0; k < *n_cigar; ++k) {
			int op, len;
			cigar = (uint32_t*)str.s;
			op = cigar[k] & 0xf; len = cigar[k] >> 4;
			if (op == 0) {
				for (i = 0; i < len; ++i) {
					if (query[x + i] != rseq[y + i]) { s_putl(&str, u); s_putc(&str, int2base[rseq[y+i]]); ++n_mm; u = 0; }
					else ++u;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < *n_cigar - 1) {
					s_putl(&str, u); s_putc(&str, '^');
					for (i = 0; i < len; ++i) s_putc(&str, int2base[rseq[y+i]]);
					u = 0; n_gap += len;
				}
				y += len;
			} else if (op == 1) x += len, n_gap += len;
		}
		s_putl(&str, u); s_need(&str, 1); str.s[str.l++] = 0;
		*NM = n_mm + n_gap;
		cigar = (uint32_t*)str.s;
	}
	if (rb >= l_pac)
		for (i = 0; i < l_query >> 1; ++i) tmp = query[i], query[i] = query[l_query-1-i], query[l_query-1-i] = tmp;
	free(rseq);
	return cigar;
}

static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = ((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

orc_aln_t orc_mem_reg2aln(const orc_opt_t *opt, const orc_idx_t *idx, int l_query, const uint8_t *query_, const orc_alnreg_t *ar)
{	/* upstream mem_reg2aln */
	orc_aln_t a;
	const orc_bns_t *bns = idx->bns;
	int i, w2, tmp, qb, qe, NM, score, is_rev, last_sc = -(1 << 30), l_MD;
	int64_t pos, rb, re;
	uint8_t *query;
	memset(&a, 0, sizeof(a));
	if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	qb = ar->qb; qe = ar->qe; rb = ar->rb; re = ar->re;
	query = malloc(l_query);
	memcpy(query, query_, l_query);
	a.mapq = ar->secondary < 0 ? orc_mem_approx_mapq_se(opt, ar) : 0;
	if (ar->secondary >= 0) a.flag |= 0x100;
	tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_del, opt->e_del);
	w2  = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_ins, opt->e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt->w) w2 = w2 < ar->w ? w2 : ar->w;
	i = 0; a.cigar = 0;
	do {
		free(a.cigar);
		w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
		a.cigar = orc_gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w2, bns->l_pac, idx->pac, qe - qb, &query[qb], rb, re, &score, &a.n_cigar, &NM);
		if (score == last_sc || w2 == opt->w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar->truesc - opt->a);
	l_MD = (int)strlen((char*)(a.cigar + a.n_cigar)) + 1;
	a.NM = NM;
	pos = orc_bns_depos(bns, rb < bns->l_pac ? rb : re - 1, &is_rev);
	a.is_rev = is_rev;
	if (a.n_cigar > 0) { /* squeeze out a leading or trailing deletion */
		if ((a.cigar[0] & 0xf) == 2) {
			pos += a.cigar[0] >> 4;
			--a.n_cigar;
			memmove(a.cigar, a.cigar + 1, a.n_cigar * 4 + l_MD);
		} else if ((a.cigar[a.n_cigar-1] & 0xf) == 2) {
			--a.n_cigar;
			memmove(a.cigar + a.n_cigar, a.cigar + a.n_cigar + 1, l_MD);
		}
	}
	if (qb != 0 || qe != l_query) { /* clipping */
		int clip5, clip3;
		clip5 = is_rev ? l_query - qe : qb;
		clip3 = is_rev ? qb : l_query - qe;
		a.cigar = realloc(a.cigar, 4 * (a.n_cigar + 2) + l_MD);
		if (clip5) {
			memmove(a.cigar + 1, a.cigar, a.n_cigar * 4 + l_MD);
			a.cigar[0] = clip5 << 4 | 3;
			++a.n_cigar;
		}
		if (clip3) {
			memmove(a.cigar + a.n_cigar + 1, a.cigar + a.n_cigar, l_MD);
			a.cigar[a.n_cigar++] = clip3 << 4 | 3;
		}
	}
	a.rid = orc_bns_pos2rid(bns, pos);
	assert(a.rid == ar->rid);
	a.pos = pos - bns->anns[a.rid].offset;
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
	free(query);
	return a;
}

/* ---------- XA ---------- */
static inline int get_pri_idx(double XA_drop_ratio, const orc_alnreg_t *a, int i)
{
	int k = a[i].secondary_all;
	if (k >= 0 && a[i].score >= a[k].score * XA_drop_ratio) return k;
	return -1;
}
static char **gen_alt(const orc_opt_t *opt, const orc_idx_t *idx, const orc_alnreg_v *a, int l_query, const uint8_t *query)
{	/* upstream mem_gen_alt */
	int i, k, r, *cnt, tot;
	str_t *aln = 0, str = {0,0,0};
	char **XA = 0;
	cnt = calloc(a->n + 1, sizeof(int));
	for (i = 0, tot = 0; i < (int)a->n; ++i) {
		r = get_pri_idx(opt->XA_drop_ratio, a->a, i);
		if (r >= 0) ++cnt[r], ++tot;
	}
	if (tot == 0) { free(cnt); return 0; }
	aln = calloc(a->n, sizeof(str_t));
	for (i = 0; i < (int)a->n; ++i) {
		orc_aln_t t;
		if ((r = get_pri_idx(opt->XA_drop_ratio, a->a, i)) < 0) continue;
		if (cnt[r] > opt->max_XA_hits_alt || cnt[r] > opt->max_XA_hits) continue;
		t = orc_mem_reg2aln(opt, idx, l_query, query, &a->a[i]);
		str.l = 0;
		s_puts(&str, idx->bns->anns[t.rid].name);
		s_putc(&str, ','); s_putc(&str, "+-"[t.is_rev]); s_putl(&str, t.pos + 1);
		s_putc(&str, ',');
		for (k = 0; k < t.n_cigar; ++k) { s_putl(&str, t.cigar[k] >> 4); s_putc(&str, "MIDSHN"[t.cigar[k] & 0xf]); }
		s_putc(&str, ','); s_putl(&str, t.NM);
		s_putc(&str, ';');
		free(t.cigar);
		s_putsn(&aln[r], str.s, str.l);
	}
	XA = calloc(a->n, sizeof(char*));
	for (k = 0; k < (int)a->n; ++k) XA[k] = aln[k].s;
	free(cnt); free(aln); free(str.s);
	return XA;
}

/* ---------- SAM ---------- */
static inline int get_rlen(int n_cigar, const uint32_t *cigar)
{
	int k, l;
	for (k = l = 0; k < n_cigar; ++k) { int op = cigar[k] & 0xf; if (op == 0 || op == 2) l += cigar[k] >> 4; }
	return l;
}

static void aln2sam(const orc_opt_t *opt, const orc_bns_t *bns, str_t *str, orc_read_t *s, int n, const orc_aln_t *list, int which, const orc_aln_t *m_, const char *rg_id)
{	/* upstream mem_aln2sam */
	int i;
	orc_aln_t ptmp = list[which], *p = &ptmp, mtmp, *m = 0;
	(void)opt;
	if (m_) mtmp = *m_, m = &mtmp;
	p->flag |= m ? 0x1 : 0;
	p->flag |= p->rid < 0 ? 0x4 : 0;
	p->flag |= m && m->rid < 0 ? 0x8 : 0;
	if (p->rid < 0 && m && m->rid >= 0) p->rid = m->rid, p->pos = m->pos, p->is_rev = m->is_rev, p->n_cigar = 0;
	if (m && m->rid < 0 && p->rid >= 0) m->rid = p->rid, m->pos = p->pos, m->is_rev = p->is_rev, m->n_cigar = 0;
	p->flag |= p->is_rev ? 0x10 : 0;
	p->flag |= m && m->is_rev ? 0x20 : 0;
	s_puts(str, s->name); s_putc(str, '\t');
	s_putl(str, (p->flag & 0xffff) | (p->flag & 0x10000 ? 0x100 : 0)); s_putc(str, '\t');
	if (p->rid >= 0) {
		s_puts(str, bns->anns[p->rid].name); s_putc(str, '\t');
		s_putl(str, p->pos + 1); s_putc(str, '\t');
		s_putl(str, p->mapq); s_putc(str, '\t');
		if (p->n_cigar) {
			for (i = 0; i < p->n_cigar; ++i) {
				int c = p->cigar[i] & 0xf;
				if (!(opt->flag & ORC_F_SOFTCLIP) && !p->is_alt && (c == 3 || c == 4)) c = which ? 4 : 3; /* hard clip supplementary */
				s_putl(str, p->cigar[i] >> 4); s_putc(str, "MIDSH"[c]);
			}
		} else s_putc(str, '*');
	} else s_putsn(str, "*\t0\t0\t*", 7);
	s_putc(str, '\t');
	if (m && m->rid >= 0) {
		if (p->rid == m->rid) s_putc(str, '=');
		else s_puts(str, bns->anns[m->rid].name);
		s_putc(str, '\t');
		s_putl(str, m->pos + 1); s_putc(str, '\t');
		if (p->rid == m->rid) {
			int64_t p0 = p->pos + (p->is_rev ? get_rlen(p->n_cigar, p->cigar) - 1 : 0);
			int64_t p1 = m->pos + (m->is_rev ? get_rlen(m->n_cigar, m->cigar) - 1 : 0);
			if (m->n_cigar == 0 || p->n_cigar == 0) s_putc(str, '0');
			else s_putl(str, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else s_putc(str, '0');
	} else s_putsn(str, "*\t0\t0", 5);
	s_putc(str, '\t');
	if (p->flag & 0x100) s_putsn(str, "*\t*", 3);
	else if (!p->is_rev) {
		int qb = 0, qe = s->l_seq;
		if (p->n_cigar && which && !(opt->flag & ORC_F_SOFTCLIP) && !p->is_alt) {
			if ((p->cigar[0] & 0xf) == 4 || (p->cigar[0] & 0xf) == 3) qb += p->cigar[0] >> 4;
			if ((p->cigar[p->n_cigar-1] & 0xf) == 4 || (p->cigar[p->n_cigar-1] & 0xf) == 3) qe -= p->cigar[p->n_cigar-1] >> 4;
		}
		for (i = qb; i < qe; ++i) s_putc(str, "ACGTN"[(int)s->seq[i]]);
		s_putc(str, '\t');
		if (s->qual) for (i = qb; i < qe; ++i) s_putc(str, s->qual[i]);
		else s_putc(str, '*');
	} else {
		int qb = 0, qe = s->l_seq;
		if (p->n_cigar && which && !(opt->flag & ORC_F_SOFTCLIP) && !p->is_alt) {
			if ((p->cigar[0] & 0xf) == 4 || (p->cigar[0] & 0xf) == 3) qe -= p->cigar[0] >> 4;
			if ((p->cigar[p->n_cigar-1] & 0xf) == 4 || (p->cigar[p->n_cigar-1] & 0xf) == 3) qb += p->cigar[p->n_cigar-1] >> 4;
		}
		for (i = qe - 1; i >= qb; --i) s_putc(str, "TGCAN"[(int)s->seq[i]]);
		s_putc(str, '\t');
		if (s->qual) for (i = qe - 1; i >= qb; --i) s_putc(str, s->qual[i]);
		else s_putc(str, '*');
	}
	if (p->n_cigar) {
		s_putsn(str, "\tNM:i:", 6); s_putl(str, p->NM);
		s_putsn(str, "\tMD:Z:", 6); s_puts(str, (char*)(p->cigar + p->n_cigar));
	}
	if (p->score >= 0) { s_putsn(str, "\tAS:i:", 6); s_putl(str, p->score); }
	if (p->sub >= 0) { s_putsn(str, "\tXS:i:", 6); s_putl(str, p->sub); }
	if (rg_id && rg_id[0]) { s_putsn(str, "\tRG:Z:", 6); s_puts(str, rg_id); }
	if (!(p->flag & 0x100)) {
		for (i = 0; i < n; ++i) if (i != which && !(list[i].flag & 0x100)) break;
		if (i < n) {
			s_putsn(str, "\tSA:Z:", 6);
			for (i = 0; i < n; ++i) {
				const orc_aln_t *r = &list[i]; int k;
				if (i == which || (r->flag & 0x100)) continue;
				s_puts(str, bns->anns[r->rid].name); s_putc(str, ',');
				s_putl(str, r->pos + 1); s_putc(str, ',');
				s_putc(str, "+-"[r->is_rev]); s_putc(str, ',');
				for (k = 0; k < r->n_cigar; ++k) { s_putl(str, r->cigar[k] >> 4); s_putc(str, "MIDSH"[r->cigar[k] & 0xf]); }
				s_putc(str, ','); s_putl(str, r->mapq);
				s_putc(str, ','); s_putl(str, r->NM);
				s_putc(str, ';');
			}
		}
	}
	if (p->XA) { s_putsn(str, "\tXA:Z:", 6); s_puts(str, p->XA); }
	if (s->comment) { s_putc(str, '\t'); s_puts(str, s->comment); }
	s_putc(str, '\n');
}

static void reg2sam(const orc_opt_t *opt, const orc_idx_t *idx, orc_read_t *s, orc_alnreg_v *a, int extra_flag, const orc_aln_t *m, const char *rg_id)
{	/* upstream mem_reg2sam */
	str_t str = {0,0,0};
	struct { size_t n, m; orc_aln_t *a; } aa = {0,0,0};
	int k, l;
	char **XA = gen_alt(opt, idx, a, s->l_seq, s->seq);
	for (k = l = 0; k < (int)a->n; ++k) {
		orc_alnreg_t *p = &a->a[k];
		orc_aln_t q;
		if (p->score < opt->T) continue;
		if (p->secondary >= 0) continue; /* no -a */
		q = orc_mem_reg2aln(opt, idx, s->l_seq, s->seq, p);
		q.XA = XA ? XA[k] : 0;
		q.flag |= extra_flag;
		if (l && p->secondary < 0) q.flag |= (opt->flag & ORC_F_NO_MULTI) ? 0x10000 : 0x800;
		if (l && !p->is_alt && q.mapq > aa.a[0].mapq) q.mapq = aa.a[0].mapq;
		PUSH(aa, orc_aln_t, q);
		++l;
	}
	if (aa.n == 0) {
		orc_aln_t t = orc_mem_reg2aln(opt, idx, s->l_seq, s->seq, 0);
		t.flag |= extra_flag;
		aln2sam(opt, idx->bns, &str, s, 1, &t, 0, m, rg_id);
	} else {
		for (k = 0; k < (int)aa.n; ++k) aln2sam(opt, idx->bns, &str, s, (int)aa.n, aa.a, k, m, rg_id);
		for (k = 0; k < (int)aa.n; ++k) free(aa.a[k].cigar);
		free(aa.a);
	}
	s->sam = str.s;
	if (XA) { for (k = 0; k < (int)a->n; ++k) free(XA[k]); free(XA); }
}

#define raw_mapq(diff, a) ((int)(6.02 * (diff) / (a) + .499))

int orc_mem_sam_pe(const orc_opt_t *opt, const orc_idx_t *idx, const orc_pestat_t pes[4], uint64_t id, orc_read_t s[2], orc_alnreg_v a[2], const char *rg_id)
{	/* upstream mem_sam_pe */
	int n = 0, i, j, z[2], o, subo, n_sub, extra_flag = 1, n_pri[2];
	str_t str = {0,0,0};
	orc_aln_t h[2];
	memset(h, 0, sizeof(h));
	if (!(opt->flag & ORC_F_NO_RESCUE)) {	/* mate rescue for the best hits */
		orc_alnreg_v b[2] = {{0,0,0},{0,0,0}};
		for (i = 0; i < 2; ++i)
			for (j = 0; j < (int)a[i].n; ++j)
				if (a[i].a[j].score >= a[i].a[0].score - opt->pen_unpaired) PUSH(b[i], orc_alnreg_t, a[i].a[j]);
		for (i = 0; i < 2; ++i)
			for (j = 0; j < (int)b[i].n && j < opt->max_matesw; ++j)
				n += orc_mem_matesw(opt, idx, pes, &b[i].a[j], s[!i].l_seq, s[!i].seq, &a[!i]);
		free(b[0].a); free(b[1].a);
	}
	n_pri[0] = orc_mem_mark_primary_se(opt, (int)a[0].n, a[0].a, id << 1 | 0);
	n_pri[1] = orc_mem_mark_primary_se(opt, (int)a[1].n, a[1].a, id << 1 | 1);
	if (opt->flag & ORC_F_NOPAIRING) goto no_pairing;
	if (n_pri[0] && n_pri[1] && (o = orc_mem_pair(opt, idx, pes, a, (int)id, &subo, &n_sub, z, n_pri)) > 0) {
		int is_multi[2], q_pe, score_un, q_se[2];
		char **XA[2];
		for (i = 0; i < 2; ++i) {
			for (j = 1; j < n_pri[i]; ++j)
				if (a[i].a[j].secondary < 0 && a[i].a[j].score >= opt->T) break;
			is_multi[i] = j < n_pri[i] ? 1 : 0;
		}
		if (is_multi[0] || is_multi[1]) goto no_pairing;
		score_un = a[0].a[0].score + a[1].a[0].score - opt->pen_unpaired;
		subo = subo > score_un ? subo : score_un;
		q_pe = raw_mapq(o - subo, opt->a);
		if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
		if (q_pe < 0) q_pe = 0;
		if (q_pe > 60) q_pe = 60;
		q_pe = (int)(q_pe * (1. - .5 * (a[0].a[0].frac_rep + a[1].a[0].frac_rep)) + .499);
		if (o > score_un) { /* paired alignment preferred */
			orc_alnreg_t *c[2] = { &a[0].a[z[0]], &a[1].a[z[1]] };
			for (i = 0; i < 2; ++i) {
				if (c[i]->secondary >= 0) c[i]->sub = a[i].a[c[i]->secondary].score, c[i]->secondary = -2;
				q_se[i] = orc_mem_approx_mapq_se(opt, c[i]);
			}
			q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
			q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
			extra_flag |= 2;
			q_se[0] = q_se[0] < raw_mapq(c[0]->score - c[0]->csub, opt->a) ? q_se[0] : raw_mapq(c[0]->score - c[0]->csub, opt->a);
			q_se[1] = q_se[1] < raw_mapq(c[1]->score - c[1]->csub, opt->a) ? q_se[1] : raw_mapq(c[1]->score - c[1]->csub, opt->a);
		} else {
			z[0] = z[1] = 0;
			q_se[0] = orc_mem_approx_mapq_se(opt, &a[0].a[0]);
			q_se[1] = orc_mem_approx_mapq_se(opt, &a[1].a[0]);
		}
		for (i = 0; i < 2; ++i) {
			int k = a[i].a[z[i]].secondary_all;
			if (k >= 0 && k < n_pri[i]) { /* swap primary and secondary */
				for (j = 0; j < (int)a[i].n; ++j)
					if (a[i].a[j].secondary_all == k || j == k) a[i].a[j].secondary_all = z[i];
				a[i].a[z[i]].secondary_all = -1;
			}
		}
		for (i = 0; i < 2; ++i) XA[i] = gen_alt(opt, idx, &a[i], s[i].l_seq, s[i].seq);
		for (i = 0; i < 2; ++i) {
			h[i] = orc_mem_reg2aln(opt, idx, s[i].l_seq, s[i].seq, &a[i].a[z[i]]);
			h[i].mapq = q_se[i];
			h[i].flag |= 0x40 << i | extra_flag;
			h[i].XA = XA[i] ? XA[i][z[i]] : 0;
		}
		aln2sam(opt, idx->bns, &str, &s[0], 1, &h[0], 0, &h[1], rg_id);
		s[0].sam = str.s; str.l = str.m = 0; str.s = 0;
		aln2sam(opt, idx->bns, &str, &s[1], 1, &h[1], 0, &h[0], rg_id);
		s[1].sam = str.s;
		for (i = 0; i < 2; ++i) {
			free(h[i].cigar);
			if (XA[i]) { for (j = 0; j < (int)a[i].n; ++j) free(XA[i][j]); free(XA[i]); }
		}
		return n;
	}
no_pairing:
	for (i = 0; i < 2; ++i) {
		int which = -1;
		if (a[i].n) {
			if (a[i].a[0].score >= opt->T) which = 0;
			else if (n_pri[i] < (int)a[i].n && a[i].a[n_pri[i]].score >= opt->T) which = n_pri[i];
		}
		if (which >= 0) h[i] = orc_mem_reg2aln(opt, idx, s[i].l_seq, s[i].seq, &a[i].a[which]);
		else h[i] = orc_mem_reg2aln(opt, idx, s[i].l_seq, s[i].seq, 0);
	}
	if (h[0].rid == h[1].rid && h[0].rid >= 0) { /* top hits form a proper pair? */
		int64_t dist; int d;
		d = infer_dir(idx->bns->l_pac, a[0].a[0].rb, a[1].a[0].rb, &dist);
		if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
	}
	reg2sam(opt, idx, &s[0], &a[0], 0x41 | extra_flag, &h[1], rg_id);
	reg2sam(opt, idx, &s[1], &a[1], 0x81 | extra_flag, &h[0], rg_id);
	free(h[0].cigar); free(h[1].cigar);
	return n;
}

/* ---------- batch driver (upstream mem_process_seqs, PE) ---------- */
typedef struct {
	const orc_opt_t *opt; const orc_idx_t *idx; int64_t n_processed; int n; orc_read_t *s; orc_alnreg_v *regs;
	const orc_pestat_t *pes; const char *rg_id; int tid, nt, stage;
} wk_t;
static void *worker(void *d)
{
	wk_t *w = d;
	if (w->stage == 1) {
		for (int i = w->tid; i < w->n; i += w->nt)
			w->regs[i] = orc_mem_align1_core(w->opt, w->idx, w->s[i].l_seq, w->s[i].seq);
	} else if (w->stage == 3) {   /* upstream worker2 without MEM_F_PE */
		for (int i = w->tid; i < w->n; i += w->nt) {
			orc_mem_mark_primary_se(w->opt, (int)w->regs[i].n, w->regs[i].a, w->n_processed + i);
			reg2sam(w->opt, w->idx, &w->s[i], &w->regs[i], 0, 0, w->rg_id);
			free(w->regs[i].a);
		}
	} else {
		for (int i = w->tid; i < w->n >> 1; i += w->nt) {
			orc_mem_sam_pe(w->opt, w->idx, w->pes, (w->n_processed >> 1) + i, &w->s[i<<1], &w->regs[i<<1], w->rg_id);
			free(w->regs[i<<1|0].a); free(w->regs[i<<1|1].a);
		}
	}
	orc_cnt_flush();
	return 0;
}
static void run_stage(wk_t *proto, int stage, int nt)
{
	/* glibc gives freed memory back to the kernel whenever the top of a thread's heap exceeds 128 KB (heap trim -> madvise / munmap under
	 * the process-wide mm lock): with the many short-lived buffers of the alignment code that serialised the workers (8 threads: 1.5x).
	 * Keep freed memory in the heaps instead. */
	static int tuned;
	if (!tuned) { tuned = 1; mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 64 << 20); mallopt(M_MMAP_THRESHOLD, 32 << 20); }
	pthread_t *th = malloc(nt * sizeof(pthread_t)); wk_t *w = malloc(nt * sizeof(wk_t));
	for (int t = 0; t < nt; ++t) { w[t] = *proto; w[t].tid = t; w[t].nt = nt; w[t].stage = stage; }
	if (nt == 1) worker(&w[0]);
	else { for (int t = 0; t < nt; ++t) pthread_create(&th[t], 0, worker, &w[t]); for (int t = 0; t < nt; ++t) pthread_join(th[t], 0); }
	free(th); free(w);
}

void orc_mem_process_pairs(const orc_opt_t *opt, const orc_idx_t *idx, int64_t n_processed, int n, orc_read_t *s,
                           const orc_pestat_t *pes0, const char *rg_id, orc_pestat_t pes_out[4], int n_threads)
{
	orc_pestat_t pes[4];
	wk_t w = { opt, idx, n_processed, n, s, calloc(n, sizeof(orc_alnreg_v)), pes, rg_id, 0, 1, 0 };
	if (n_threads < 1) n_threads = 1;
	struct timespec t0_, t1_, t2_; clock_gettime(CLOCK_MONOTONIC, &t0_);
	run_stage(&w, 1, n_threads);
	clock_gettime(CLOCK_MONOTONIC, &t1_);
	if (pes0) memcpy(pes, pes0, sizeof(pes));
	else orc_mem_pestat(opt, idx->bns->l_pac, n, w.regs, pes);
	if (pes_out) memcpy(pes_out, pes, sizeof(pes));
	run_stage(&w, 2, n_threads);
	clock_gettime(CLOCK_MONOTONIC, &t2_);
	if (getenv("ORC_TIMING")) fprintf(stderr, "[orc] stage 1 %.3f s, stage 2 %.3f s (%d threads)\n", (t1_.tv_sec - t0_.tv_sec) + 1e-9 * (t1_.tv_nsec - t0_.tv_nsec), (t2_.tv_sec - t1_.tv_sec) + 1e-9 * (t2_.tv_nsec - t1_.tv_nsec), n_threads);
	free(w.regs);
}

void orc_mem_process_reads(const orc_opt_t *opt, const orc_idx_t *idx, int64_t n_processed, int n, orc_read_t *s, const char *rg_id, int n_threads)
{	/* upstream mem_process_seqs without MEM_F_PE: every read on its own (single-end input, and the unpaired reads of a `-p` stream) */
	wk_t w = { opt, idx, n_processed, n, s, calloc(n, sizeof(orc_alnreg_v)), 0, rg_id, 0, 1, 0 };
	if (n_threads < 1) n_threads = 1;
	run_stage(&w, 1, n_threads);
	run_stage(&w, 3, n_threads);
	free(w.regs);
}

char *orc_sam_header(const orc_idx_t *idx, const char *rg_line, const char *pg_cl)
{	/* upstream bwa_print_sam_hdr + the @PG line of main() */
	str_t s = {0,0,0};
	for (int i = 0; i < idx->bns->n_seqs; ++i) {
		s_puts(&s, "@SQ\tSN:"); s_puts(&s, idx->bns->anns[i].name); s_puts(&s, "\tLN:"); s_putl(&s, idx->bns->anns[i].len); s_putc(&s, '\n');
	}
	if (rg_line && rg_line[0]) { s_puts(&s, rg_line); s_putc(&s, '\n'); }
	if (pg_cl) { s_puts(&s, "@PG\tID:bwa\tPN:bwa\tVN:0.7.12-ssgpu\tCL:"); s_puts(&s, pg_cl); s_putc(&s, '\n'); }
	return s.s;
}
