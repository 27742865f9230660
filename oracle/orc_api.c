/*
 * oracle/orc_api.c -- CPU ORACLE (test infrastructure): flat, ctypes-friendly wrappers used by
 * tests/ to compare the HIP path with the oracle stage by stage.
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

typedef struct {
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp;
	float frac_rep;
	uint64_t hash;
} orc_flatreg_t;

static void flatten(const orc_alnreg_t *a, orc_flatreg_t *f)
{
	f->rb = a->rb; f->re = a->re; f->qb = a->qb; f->qe = a->qe; f->rid = a->rid; f->score = a->score; f->truesc = a->truesc;
	f->sub = a->sub; f->alt_sc = a->alt_sc; f->csub = a->csub; f->sub_n = a->sub_n; f->w = a->w; f->seedcov = a->seedcov;
	f->secondary = a->secondary; f->secondary_all = a->secondary_all; f->seedlen0 = a->seedlen0; f->n_comp = a->n_comp;
	f->frac_rep = a->frac_rep; f->hash = a->hash;
}

orc_opt_t *orc_api_opt_new(void) { orc_opt_t *o = malloc(sizeof(orc_opt_t)); orc_opt_init(o); return o; }
/* scoring of the stage-level checks (bwa mem -A -B -O -E): a, b, gap open / extend for deletions and insertions */
void orc_api_opt_scores(orc_opt_t *o, int a, int b, int o_del, int e_del, int o_ins, int e_ins)
{ o->a = a; o->b = b; o->o_del = o_del; o->e_del = e_del; o->o_ins = o_ins; o->e_ins = e_ins; orc_fill_scmat(o); }
/* the chain filter's knobs (bwa mem -D, -W, -N and the compiled-in mask_level / max_chain_gap), for the chaining kernels' tests */
void orc_api_opt_chain(orc_opt_t *o, float drop_ratio, float mask_level, int min_chain_weight, int max_chain_extend, int max_chain_gap)
{ o->drop_ratio = drop_ratio; o->mask_level = mask_level; o->min_chain_weight = min_chain_weight; o->max_chain_extend = max_chain_extend; o->max_chain_gap = max_chain_gap; }
void orc_api_free(void *p) { free(p); }

/* mem_collect_intv for one read; returns the number of intervals (out may be smaller than needed) */
int orc_api_collect_intv(const orc_opt_t *opt, const orc_idx_t *idx, int len, const uint8_t *seq, orc_intv_t *out, int cap)
{
	orc_intv_v mem = {0,0,0};
	orc_collect_intv(opt, idx->bwt, len, seq, &mem);
	int n = (int)mem.n;
	memcpy(out, mem.a, sizeof(orc_intv_t) * (n < cap ? n : cap));
	free(mem.a);
	return n;
}

/* the seeds mem_chain() visits for one read, in its order, before chaining (upstream bwamem.c mem_chain: for every interval of mem_collect_intv,
 * step = x[2] > max_occ ? x[2] / max_occ : 1, occurrences k = 0, step, ... while count < max_occ; rbeg = bwt_sa(x[0] + k); rid = bns_intv2rid,
 * seeds with rid < 0 are dropped by upstream and reported here with their rid).  out: 4 values per seed (rbeg, qbeg, len, rid).  Returns the count. */
int64_t orc_api_seeds(const orc_opt_t *opt, const orc_idx_t *idx, int len, const uint8_t *seq, int64_t *out, int64_t cap)
{
	orc_intv_v mem = {0,0,0};
	int64_t n = 0;
	if (len < opt->min_seed_len) return 0;
	orc_collect_intv(opt, idx->bwt, len, seq, &mem);
	for (size_t i = 0; i < mem.n; ++i) {
		const orc_intv_t *p = &mem.a[i];
		const int slen = (int)((uint32_t)p->info - (uint32_t)(p->info >> 32));
		const int64_t step = p->x[2] > (uint64_t)opt->max_occ ? (int64_t)(p->x[2] / opt->max_occ) : 1;
		int count = 0;
		for (int64_t k = 0; k < (int64_t)p->x[2] && count < opt->max_occ; k += step, ++count, ++n) {
			if (n >= cap) continue;
			const int64_t rbeg = (int64_t)orc_bwt_sa(idx->bwt, p->x[0] + k);
			out[4 * n] = rbeg; out[4 * n + 1] = (int64_t)(p->info >> 32); out[4 * n + 2] = slen; out[4 * n + 3] = orc_bns_intv2rid(idx->bns, rbeg, rbeg + slen);
		}
	}
	free(mem.a);
	return n;
}

/* mem_align1_core for a batch of reads; regs are appended to out (cap entries); reg_off[n+1] */
int64_t orc_api_align1_batch(const orc_opt_t *opt, const orc_idx_t *idx, int n_reads, const uint8_t *seq, const int64_t *off,
                             int64_t *reg_off, orc_flatreg_t *out, int64_t cap)
{
	int64_t tot = 0;
	for (int r = 0; r < n_reads; ++r) {
		int len = (int)(off[r+1] - off[r]);
		uint8_t *s = malloc(len + 1); memcpy(s, seq + off[r], len);
		orc_alnreg_v v = orc_mem_align1_core(opt, idx, len, s);
		reg_off[r] = tot;
		for (size_t i = 0; i < v.n; ++i) { if (tot < cap) flatten(&v.a[i], &out[tot]); ++tot; }
		free(v.a); free(s);
	}
	reg_off[n_reads] = tot;
	return tot;
}

/* ksw_global2 with the CIGAR copied out */
int orc_api_global2(const orc_opt_t *opt, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w, int *n_cigar, uint32_t *cigar, int cap)
{
	uint32_t *cg = 0; int n = 0;
	int sc = orc_ksw_global2(qlen, q, tlen, t, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, &n, &cg);
	*n_cigar = n;
	memcpy(cigar, cg, 4 * (n < cap ? n : cap));
	free(cg);
	return sc;
}

void orc_api_extend2(const orc_opt_t *opt, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w, int end_bonus, int zdrop, int h0, int out[6])
{
	out[0] = orc_ksw_extend2(qlen, q, tlen, t, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, end_bonus, zdrop, h0,
	                         &out[1], &out[2], &out[3], &out[4], &out[5]);
}

void orc_api_align2(const orc_opt_t *opt, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int xtra, int out[7])
{
	uint8_t *qc = malloc(qlen + 1), *tc = malloc(tlen + 1);
	memcpy(qc, q, qlen); memcpy(tc, t, tlen);
	orc_kswr_t r = orc_ksw_align2(qlen, qc, tlen, tc, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, xtra);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
	free(qc); free(tc);
}

/* mem_process_seqs (PE) for one upstream batch; returns a malloc'd buffer with the SAM lines of all
 * reads in input order and fills sam_off[2*n_pairs+1]; pes_out receives the 4 insert-size models */
char *orc_api_process_pairs(const orc_opt_t *opt, const orc_idx_t *idx, int n_pairs, const uint8_t *seq, const int64_t *off,
                            const char **names, const char **quals, int64_t n_processed, const char *rg_id, int n_threads,
                            const orc_pestat_t *pes0, orc_pestat_t *pes_out, int64_t *sam_off)
{
	int n = 2 * n_pairs;
	orc_read_t *s = calloc(n, sizeof(orc_read_t));
	for (int i = 0; i < n; ++i) {
		s[i].l_seq = (int)(off[i+1] - off[i]);
		s[i].seq = malloc(s[i].l_seq + 1); memcpy(s[i].seq, seq + off[i], s[i].l_seq);
		s[i].name = (char*)names[i]; s[i].qual = quals ? (char*)quals[i] : 0; s[i].comment = 0;
	}
	orc_mem_process_pairs(opt, idx, n_processed, n, s, pes0, rg_id, pes_out, n_threads);
	size_t tot = 0;
	for (int i = 0; i < n; ++i) tot += strlen(s[i].sam);
	char *buf = malloc(tot + 1); size_t l = 0;
	for (int i = 0; i < n; ++i) { size_t k = strlen(s[i].sam); sam_off[i] = (int64_t)l; memcpy(buf + l, s[i].sam, k); l += k; free(s[i].sam); free(s[i].seq); }
	sam_off[n] = (int64_t)l; buf[l] = 0;
	free(s);
	return buf;
}

/* Exposure of the one container that is not upstream's (orc_mem.c: the chains of a read in a position-sorted array instead of klib's B-tree): every read through
 * both, on n_threads threads.  out[0] = reads whose chain lists differ, [1] = reads with more than 9 chains AND two chains at one position (the only reads that can
 * differ), [2] = reads with more than 9 chains, [3] = reads with two chains at one position, [4] = reads looked at. */
#include <pthread.h>
typedef struct { const orc_opt_t *opt; const orc_idx_t *idx; const uint8_t *seq; const int64_t *off; int r0, r1; int64_t c[4]; int32_t *which; } expo_t;
static void *expo_worker(void *p)
{
	expo_t *w = (expo_t*)p;
	for (int r = w->r0; r < w->r1; ++r) {
		int nc = 0, dup = 0;
		const int d = orc_chain_exposure(w->opt, w->idx, (int)(w->off[r + 1] - w->off[r]), w->seq + w->off[r], &nc, &dup);
		w->c[0] += d; w->c[1] += nc > 9 && dup; w->c[2] += nc > 9; w->c[3] += dup;
		if (w->which) w->which[r] = (d ? 1 : 0) | (nc > 9 && dup ? 2 : 0);
	}
	return 0;
}
void orc_api_chain_exposure(const orc_opt_t *opt, const orc_idx_t *idx, int n_reads, const uint8_t *seq, const int64_t *off, int n_threads, int64_t out[5], int32_t *which /* may be NULL: per read, bit 0 differs, bit 1 > 9 chains and a shared position */)
{
	if (n_threads < 1) n_threads = 1;
	if (n_threads > n_reads) n_threads = n_reads > 0 ? n_reads : 1;
	pthread_t *th = malloc(n_threads * sizeof(pthread_t)); expo_t *w = calloc(n_threads, sizeof(expo_t));
	for (int t = 0; t < n_threads; ++t) {
		w[t].opt = opt; w[t].idx = idx; w[t].seq = seq; w[t].off = off; w[t].which = which;
		w[t].r0 = (int)((int64_t)n_reads * t / n_threads); w[t].r1 = (int)((int64_t)n_reads * (t + 1) / n_threads);
		pthread_create(&th[t], 0, expo_worker, &w[t]);
	}
	out[0] = out[1] = out[2] = out[3] = 0; out[4] = n_reads;
	for (int t = 0; t < n_threads; ++t) { pthread_join(th[t], 0); for (int k = 0; k < 4; ++k) out[k] += w[t].c[k]; }
	free(th); free(w);
}
/* the container of mem_chain for the calls this thread makes (orc_align1 and the batch entry points run their workers on threads of their own: see ORC_CHAIN_KBTREE) */
void orc_api_set_chain_container(int kbtree) { orc_set_chain_container(kbtree); }

/* exposure of the other shared deviation (orc_ksw.c local_sw): on = 1 makes every orc_ksw_align2 call run a second time with E never opened from an F-raised H
 * and counts the calls whose result changes; out[0] = calls, out[1] = calls that differ (both since the process began) */
extern int orc_lazyf_count; extern uint64_t orc_lazyf_calls, orc_lazyf_differ;
void orc_api_lazyf(int on, uint64_t out[2]) { if (on >= 0) orc_lazyf_count = on; out[0] = orc_lazyf_calls; out[1] = orc_lazyf_differ; }
