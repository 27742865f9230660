/*
 * oracle/orc_mem.c -- CPU ORACLE (test infrastructure): BWA-MEM single-end core.
 *
 * Restates upstream lh3/bwa bwt.c (bwt_smem1a, bwt_seed_strategy1) and bwamem.c
 * (mem_opt_init, mem_collect_intv, mem_chain, test_and_merge, mem_chain_weight, mem_chain_flt,
 * mem_flt_chained_seeds [guard only], mem_chain2aln, mem_patch_reg, mem_sort_dedup_patch,
 * mem_align1_core).  Upstream sources are absent from /root/reference (empty submodule
 * src/bwa, .gitmodules:16-18); behaviour follows SURVEY.md Appendix B.  PARITY UNPINNED.
 *
 * Documented simplifications relative to upstream:
 *  - chains live in a position-sorted array with the klib B-tree's single-leaf semantics
 *    (lookup = first chain with equal pos, else the largest smaller one; insert right after it).
 *    This equals kbtree for all inputs without duplicate chain positions and for <=9 chains.
 *  - ALT contigs (.alt file) are not modelled (is_alt == 0 everywhere).
 *  - mem_flt_chained_seeds only runs for reads >= ~1 kb upstream; such reads are rejected.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "orc.h"

#define PUSH(v, T, x) do { if ((v).n == (v).m) { (v).m = (v).m ? (v).m << 1 : 4; (v).a = realloc((v).a, sizeof(T) * (v).m); } (v).a[(v).n++] = (x); } while (0)

void orc_opt_init(orc_opt_t *o)
{	/* upstream mem_opt_init */
	memset(o, 0, sizeof(*o));
	o->flag = 0;
	o->a = 1; o->b = 4;
	o->o_del = o->o_ins = 6;
	o->e_del = o->e_ins = 1;
	o->w = 100;
	o->T = 30;
	o->zdrop = 100;
	o->pen_unpaired = 17;
	o->pen_clip5 = o->pen_clip3 = 5;
	o->max_mem_intv = 20;
	o->min_seed_len = 19;
	o->split_width = 10;
	o->max_occ = 500;
	o->max_chain_gap = 10000;
	o->max_ins = 10000;
	o->mask_level = 0.50;
	o->drop_ratio = 0.50;
	o->XA_drop_ratio = 0.80;
	o->split_factor = 1.5;
	o->chunk_size = 10000000;
	o->n_threads = 1;
	o->max_XA_hits = 5;
	o->max_XA_hits_alt = 200;
	o->max_matesw = 50;
	o->mask_level_redun = 0.95;
	o->min_chain_weight = 0;
	o->max_chain_extend = 1<<30;
	o->mapQ_coef_len = 50; o->mapQ_coef_fac = log(o->mapQ_coef_len);
	orc_fill_scmat(o);
}

void orc_fill_scmat(orc_opt_t *o)
{	/* upstream bwa_fill_scmat */
	for (int i = 0, k = 0; i < 4; ++i) {
		for (int j = 0; j < 4; ++j) o->mat[k++] = i == j ? o->a : -o->b;
		o->mat[k++] = -1;
	}
	for (int j = 0; j < 5; ++j) o->mat[20 + j] = -1;
}

/* ---------------- SMEM ---------------- */
static inline void set_intv(const orc_bwt_t *bwt, int c, orc_intv_t *ik)
{
	ik->x[0] = bwt->L2[c] + 1; ik->x[2] = bwt->L2[c+1] - bwt->L2[c]; ik->x[1] = bwt->L2[3-c] + 1; ik->info = 0;
}
static void reverse_intvs(orc_intv_v *p)
{
	for (size_t j = 0; j < p->n >> 1; ++j) { orc_intv_t t = p->a[p->n-1-j]; p->a[p->n-1-j] = p->a[j]; p->a[j] = t; }
}

int orc_smem1(const orc_bwt_t *bwt, int len, const uint8_t *q, int x, int min_intv, uint64_t max_intv, orc_intv_v *mem, orc_intv_v tmp[2])
{	/* upstream bwt_smem1a */
	int i, j, c, ret;
	orc_intv_t ik, ok[4];
	orc_intv_v *prev, *curr, *swap;
	mem->n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	prev = &tmp[0]; curr = &tmp[1];
	set_intv(bwt, q[x], &ik);
	ik.info = x + 1;
	for (i = x + 1, curr->n = 0; i < len; ++i) { /* forward extension */
		if (ik.x[2] < max_intv) { PUSH(*curr, orc_intv_t, ik); break; }
		else if (q[i] < 4) {
			c = 3 - q[i];
			orc_bwt_extend(bwt, &ik, ok, 0);
			if (ok[c].x[2] != ik.x[2]) {
				PUSH(*curr, orc_intv_t, ik);
				if (ok[c].x[2] < (uint64_t)min_intv) break;
			}
			ik = ok[c]; ik.info = i + 1;
		} else { PUSH(*curr, orc_intv_t, ik); break; }
	}
	if (i == len) PUSH(*curr, orc_intv_t, ik);
	reverse_intvs(curr);
	ret = (int)curr->a[0].info;
	swap = curr; curr = prev; prev = swap;
	for (i = x - 1; i >= -1; --i) { /* backward extension */
		c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		for (j = 0, curr->n = 0; j < (int)prev->n; ++j) {
			orc_intv_t *p = &prev->a[j];
			if (c >= 0 && ik.x[2] >= max_intv) orc_bwt_extend(bwt, p, ok, 1);
			if (c < 0 || ik.x[2] < max_intv || ok[c].x[2] < (uint64_t)min_intv) {
				if (curr->n == 0) {
					if (mem->n == 0 || (uint64_t)(i + 1) < mem->a[mem->n-1].info >> 32) {
						ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
						PUSH(*mem, orc_intv_t, ik);
					}
				}
			} else if (curr->n == 0 || ok[c].x[2] != curr->a[curr->n-1].x[2]) {
				ok[c].info = p->info;
				PUSH(*curr, orc_intv_t, ok[c]);
			}
		}
		if (curr->n == 0) break;
		swap = curr; curr = prev; prev = swap;
	}
	reverse_intvs(mem);
	return ret;
}

int orc_seed_strategy1(const orc_bwt_t *bwt, int len, const uint8_t *q, int x, int min_len, int max_intv, orc_intv_t *mem)
{	/* upstream bwt_seed_strategy1 */
	int i, c;
	orc_intv_t ik, ok[4];
	memset(mem, 0, sizeof(orc_intv_t));
	if (q[x] > 3) return x + 1;
	set_intv(bwt, q[x], &ik);
	for (i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			orc_bwt_extend(bwt, &ik, ok, 0);
			if (ok[c].x[2] < (uint64_t)max_intv && i - x >= min_len) {
				*mem = ok[c];
				mem->info = (uint64_t)x << 32 | (i + 1);
				return i + 1;
			}
			ik = ok[c];
		} else return i + 1;
	}
	return len;
}

static int intv_lt(const void *a, const void *b) { return ((const orc_intv_t*)a)->info < ((const orc_intv_t*)b)->info; }

void orc_collect_intv(const orc_opt_t *opt, const orc_bwt_t *bwt, int len, const uint8_t *seq, orc_intv_v *mem)
{	/* upstream mem_collect_intv */
	int i, k, x = 0, old_n;
	int start_width = 1;
	int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	orc_intv_v mem1 = {0,0,0}, tmp[2] = {{0,0,0},{0,0,0}};
	mem->n = 0;
	while (x < len) { /* pass 1: all SMEMs */
		if (seq[x] < 4) {
			x = orc_smem1(bwt, len, seq, x, start_width, 0, &mem1, tmp);
			for (i = 0; i < (int)mem1.n; ++i) {
				orc_intv_t *p = &mem1.a[i];
				int slen = (uint32_t)p->info - (p->info >> 32);
				if (slen >= opt->min_seed_len) PUSH(*mem, orc_intv_t, *p);
			}
		} else ++x;
	}
	old_n = (int)mem->n; /* pass 2: re-seed long, low-occurrence SMEMs from their middle */
	for (k = 0; k < old_n; ++k) {
		orc_intv_t *p = &mem->a[k];
		int start = p->info >> 32, end = (int32_t)p->info;
		if (end - start < split_len || p->x[2] > (uint64_t)opt->split_width) continue;
		orc_smem1(bwt, len, seq, (start + end) >> 1, (int)p->x[2] + 1, 0, &mem1, tmp);
		for (i = 0; i < (int)mem1.n; ++i)
			if ((uint32_t)mem1.a[i].info - (mem1.a[i].info >> 32) >= (uint32_t)opt->min_seed_len)
				PUSH(*mem, orc_intv_t, mem1.a[i]);
	}
	if (opt->max_mem_intv > 0) { /* pass 3: LAST-like forward seeds */
		x = 0;
		while (x < len) {
			if (seq[x] < 4) {
				orc_intv_t m;
				x = orc_seed_strategy1(bwt, len, seq, x, opt->min_seed_len, (int)opt->max_mem_intv, &m);
				if (m.x[2] > 0) PUSH(*mem, orc_intv_t, m);
			} else ++x;
		}
	}
	orc_introsort(mem->a, mem->n, sizeof(orc_intv_t), intv_lt);
	free(mem1.a); free(tmp[0].a); free(tmp[1].a);
}

/* ---------------- chaining ---------------- */
static int test_and_merge(const orc_opt_t *opt, int64_t l_pac, orc_chain_t *c, const orc_seed_t *p, int seed_rid)
{	/* upstream test_and_merge */
	int64_t qend, rend, x, y;
	const orc_seed_t *last = &c->seeds[c->n-1];
	qend = last->qbeg + last->len;
	rend = last->rbeg + last->len;
	if (seed_rid != c->rid) return 0;
	if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
		return 1; /* contained */
	if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0;
	x = p->qbeg - last->qbeg;
	y = p->rbeg - last->rbeg;
	if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last->len < opt->max_chain_gap && y - last->len < opt->max_chain_gap) {
		if (c->n == c->m) { c->m <<= 1; c->seeds = realloc(c->seeds, c->m * sizeof(orc_seed_t)); }
		c->seeds[c->n++] = *p;
		return 1;
	}
	return 0;
}

/* ---- klib's B-tree as upstream mem_chain uses it (kbtree.h, KBTREE_INIT(chn, mem_chain_t, chain_cmp) with kb_init(chn, KB_DEFAULT_SIZE = 512)) ----
 * [RECALL: kbtree.h is not in the reference tree.]  Keys are compared by `pos` alone, equal keys are allowed.  For sizeof(mem_chain_t) = 40 the minimum
 * degree is t = ((512 - 4 - 8) / (8 + 40) + 1) >> 1 = 5: a node holds up to 2t - 1 = 9 keys, so a read's chains live in ONE leaf until the tenth is put, and the
 * position-sorted array below is that leaf.  From the tenth chain on kb_putp splits full nodes on its way down (__kb_split: the median key moves up) and both
 * kb_intervalp and kb_putp search node by node (__kb_getp_aux: the FIRST key >= k of the node at hand, one back when that key is greater) -- with several chains at
 * one position, which of them a lookup meets and where a new one lands then depends on the node boundaries.  orc_chain_container selects the container:
 * 1 (default) = this tree, 0 = the array (ORC_CHAIN_ARRAY in the environment; what the HIP kernels keep for every read that is not flagged: csrc/k_chain.h ssg_kbflag).  orc_chain_exposure() runs a read through both and says whether they differ. */
#define KB_T 5
typedef struct kbn { int is_internal, n; int key[2 * KB_T - 1]; struct kbn *ptr[2 * KB_T]; } kbn_t;   /* keys = indices into the pool of chains */
typedef struct { kbn_t *root; const orc_chain_t *pool; } kbt_t;
static int kb_getp_aux(const kbt_t *b, const kbn_t *x, int64_t pos, int *r)
{
	int tr, *rr = r ? r : &tr, begin = 0, end = x->n;
	if (x->n == 0) return -1;
	while (begin < end) { const int mid = (begin + end) >> 1; if (b->pool[x->key[mid]].pos < pos) begin = mid + 1; else end = mid; }
	if (begin == x->n) { *rr = 1; return x->n - 1; }
	*rr = (pos > b->pool[x->key[begin]].pos) - (pos < b->pool[x->key[begin]].pos);
	if (*rr < 0) --begin;
	return begin;
}
static int kb_interval_lower(const kbt_t *b, int64_t pos)
{	/* kb_intervalp's `lower`: index of a chain with the largest position <= pos, or -1 */
	const kbn_t *x = b->root; int lower = -1;
	while (x) {
		int r = 0; const int i = kb_getp_aux(b, x, pos, &r);
		if (i >= 0 && r == 0) return x->key[i];
		if (i >= 0) lower = x->key[i];
		if (!x->is_internal) return lower;
		x = x->ptr[i + 1];
	}
	return lower;
}
static void kb_split(kbn_t *x, int i, kbn_t *y)
{
	kbn_t *z = calloc(1, sizeof(kbn_t));
	z->is_internal = y->is_internal; z->n = KB_T - 1;
	memcpy(z->key, y->key + KB_T, sizeof(int) * (KB_T - 1));
	if (y->is_internal) memcpy(z->ptr, y->ptr + KB_T, sizeof(kbn_t*) * KB_T);
	y->n = KB_T - 1;
	memmove(x->ptr + i + 2, x->ptr + i + 1, sizeof(kbn_t*) * (x->n - i));
	x->ptr[i + 1] = z;
	memmove(x->key + i + 1, x->key + i, sizeof(int) * (x->n - i));
	x->key[i] = y->key[KB_T - 1];
	++x->n;
}
static void kb_putp_aux(const kbt_t *b, kbn_t *x, int k)
{
	const int64_t pos = b->pool[k].pos;
	if (!x->is_internal) {
		const int i = kb_getp_aux(b, x, pos, 0);
		if (i != x->n - 1) memmove(x->key + i + 2, x->key + i + 1, (x->n - i - 1) * sizeof(int));
		x->key[i + 1] = k; ++x->n;
	} else {
		int i = kb_getp_aux(b, x, pos, 0) + 1;
		if (x->ptr[i]->n == 2 * KB_T - 1) {
			kb_split(x, i, x->ptr[i]);
			if (pos > b->pool[x->key[i]].pos) ++i;
		}
		kb_putp_aux(b, x->ptr[i], k);
	}
}
static void kb_putp(kbt_t *b, int k)
{
	kbn_t *r = b->root;
	if (r->n == 2 * KB_T - 1) {
		kbn_t *s = calloc(1, sizeof(kbn_t));
		b->root = s; s->is_internal = 1; s->n = 0; s->ptr[0] = r;
		kb_split(s, 0, r);
		r = s;
	}
	kb_putp_aux(b, r, k);
}
static void kb_traverse(const kbn_t *x, int *out, int *n)
{	/* __kb_traverse: in order */
	if (!x) return;
	for (int i = 0; i < x->n; ++i) { if (x->is_internal) kb_traverse(x->ptr[i], out, n); out[(*n)++] = x->key[i]; }
	if (x->is_internal) kb_traverse(x->ptr[x->n], out, n);
}
static void kb_free(kbn_t *x) { if (!x) return; if (x->is_internal) for (int i = 0; i <= x->n; ++i) kb_free(x->ptr[i]); free(x); }
static __thread int orc_chain_container = -1;   /* -1: not chosen yet on this thread -> upstream's tree, unless ORC_CHAIN_ARRAY is set in the environment */
void orc_set_chain_container(int kbtree) { orc_chain_container = kbtree; }
static int chain_container(void) { if (orc_chain_container < 0) orc_chain_container = getenv("ORC_CHAIN_ARRAY") ? 0 : 1; return orc_chain_container; }

orc_chain_v orc_mem_chain(const orc_opt_t *opt, const orc_idx_t *idx, int len, const uint8_t *seq)
{	/* upstream mem_chain */
	int i, b, e, l_rep;
	const orc_bwt_t *bwt = idx->bwt; const orc_bns_t *bns = idx->bns;
	int64_t l_pac = bns->l_pac;
	orc_chain_v chain = {0,0,0};
	orc_intv_v mem = {0,0,0};
	kbt_t tree = { 0, 0 };
	if (len < opt->min_seed_len) return chain;
	const int use_tree = chain_container();
	if (use_tree) tree.root = calloc(1, sizeof(kbn_t));
	orc_collect_intv(opt, bwt, len, seq, &mem);
	for (i = 0, b = e = l_rep = 0; i < (int)mem.n; ++i) { /* frac_rep */
		orc_intv_t *p = &mem.a[i];
		int sb = (p->info >> 32), se = (uint32_t)p->info;
		if (p->x[2] <= (uint64_t)opt->max_occ) continue;
		if (sb > e) l_rep += e - b, b = sb, e = se;
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	for (i = 0; i < (int)mem.n; ++i) {
		orc_intv_t *p = &mem.a[i];
		int step, count, slen = (uint32_t)p->info - (p->info >> 32);
		int64_t k;
		step = p->x[2] > (uint64_t)opt->max_occ ? p->x[2] / opt->max_occ : 1;
		for (k = count = 0; k < (int64_t)p->x[2] && count < opt->max_occ; k += step, ++count) {
			orc_seed_t s;
			int rid, to_add = 0;
			s.rbeg = orc_bwt_sa(bwt, p->x[0] + k);
			s.qbeg = p->info >> 32;
			s.score = s.len = slen;
			rid = orc_bns_intv2rid(bns, s.rbeg, s.rbeg + s.len);
			if (rid < 0) continue;
			if (use_tree) {   /* klib's B-tree; chain.a is the pool, in order of creation */
				tree.pool = chain.a;
				const int lw = chain.n ? kb_interval_lower(&tree, s.rbeg) : -1;
				if (lw < 0 || !test_and_merge(opt, l_pac, &chain.a[lw], &s, rid)) {
					orc_chain_t tmp; memset(&tmp, 0, sizeof(tmp));
					tmp.pos = s.rbeg; tmp.n = 1; tmp.m = 4;
					tmp.seeds = calloc(tmp.m, sizeof(orc_seed_t));
					tmp.seeds[0] = s; tmp.rid = rid; tmp.is_alt = 0;
					if (chain.n == chain.m) { chain.m = chain.m ? chain.m << 1 : 8; chain.a = realloc(chain.a, chain.m * sizeof(orc_chain_t)); }
					chain.a[chain.n++] = tmp;
					tree.pool = chain.a;
					kb_putp(&tree, (int)chain.n - 1);
				}
				continue;
			}
			/* sorted-array stand-in for kb_intervalp/kb_putp (see file header) */
			size_t lo = 0, hi = chain.n;
			while (lo < hi) { size_t mid = (lo + hi) >> 1; if (chain.a[mid].pos < s.rbeg) lo = mid + 1; else hi = mid; }
			long li = (lo < chain.n && chain.a[lo].pos == s.rbeg) ? (long)lo : (long)lo - 1;
			if (chain.n) {
				if (li < 0 || !test_and_merge(opt, l_pac, &chain.a[li], &s, rid)) to_add = 1;
			} else to_add = 1;
			if (to_add) {
				orc_chain_t tmp; memset(&tmp, 0, sizeof(tmp));
				tmp.pos = s.rbeg; tmp.n = 1; tmp.m = 4;
				tmp.seeds = calloc(tmp.m, sizeof(orc_seed_t));
				tmp.seeds[0] = s; tmp.rid = rid; tmp.is_alt = 0;
				if (chain.n == chain.m) { chain.m = chain.m ? chain.m << 1 : 8; chain.a = realloc(chain.a, chain.m * sizeof(orc_chain_t)); }
				size_t at = (size_t)(li + 1);
				memmove(chain.a + at + 1, chain.a + at, (chain.n - at) * sizeof(orc_chain_t));
				chain.a[at] = tmp; ++chain.n;
			}
		}
	}
	if (use_tree) {   /* the chains in the tree's order (mem_chain: __kb_traverse into the vector) */
		int *ord = malloc((chain.n + 1) * sizeof(int)), no = 0;
		orc_chain_t *srt = malloc((chain.n + 1) * sizeof(orc_chain_t));
		kb_traverse(tree.root, ord, &no);
		for (i = 0; i < no; ++i) srt[i] = chain.a[ord[i]];
		memcpy(chain.a, srt, no * sizeof(orc_chain_t));
		free(ord); free(srt); kb_free(tree.root);
	}
	for (i = 0; i < (int)chain.n; ++i) chain.a[i].frac_rep = (float)l_rep / len;
	free(mem.a);
	return chain;
}

/* One read through both containers.  Returns 1 when the chain lists differ (positions, seed lists or order): the read is one on which the array -- and so the
 * HIP kernels -- may part from upstream's tree.  *n_chains = chains before filtering, *dup = some two of them share a position. */
int orc_chain_exposure(const orc_opt_t *opt, const orc_idx_t *idx, int len, const uint8_t *seq, int *n_chains, int *dup)
{
	const int saved = orc_chain_container;
	orc_chain_container = 0; orc_chain_v a = orc_mem_chain(opt, idx, len, seq);
	orc_chain_container = 1; orc_chain_v b = orc_mem_chain(opt, idx, len, seq);
	orc_chain_container = saved;
	int differ = a.n != b.n;
	for (size_t i = 0; !differ && i < a.n; ++i) {
		differ = a.a[i].pos != b.a[i].pos || a.a[i].n != b.a[i].n || a.a[i].rid != b.a[i].rid;
		for (int j = 0; !differ && j < a.a[i].n; ++j) differ = memcmp(&a.a[i].seeds[j], &b.a[i].seeds[j], sizeof(orc_seed_t)) != 0;
	}
	*n_chains = (int)a.n; *dup = 0;
	for (size_t i = 1; i < a.n; ++i) if (a.a[i].pos == a.a[i-1].pos) *dup = 1;
	for (size_t i = 0; i < a.n; ++i) free(a.a[i].seeds);
	for (size_t i = 0; i < b.n; ++i) free(b.a[i].seeds);
	free(a.a); free(b.a);
	return differ;
}

static int chain_weight(const orc_chain_t *c)
{	/* upstream mem_chain_weight */
	int64_t end; int j, w = 0, tmp;
	for (j = 0, end = 0; j < c->n; ++j) {
		const orc_seed_t *s = &c->seeds[j];
		if (s->qbeg >= end) w += s->len;
		else if (s->qbeg + s->len > end) w += s->qbeg + s->len - end;
		end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
	}
	tmp = w; w = 0;
	for (j = 0, end = 0; j < c->n; ++j) {
		const orc_seed_t *s = &c->seeds[j];
		if (s->rbeg >= end) w += s->len;
		else if (s->rbeg + s->len > end) w += s->rbeg + s->len - end;
		end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
	}
	w = w < tmp ? w : tmp;
	return w < 1<<30 ? w : (1<<30) - 1;
}

static int flt_lt(const void *a, const void *b) { return ((const orc_chain_t*)a)->w > ((const orc_chain_t*)b)->w; }
#define chn_beg(ch) ((ch).seeds->qbeg)
#define chn_end(ch) ((ch).seeds[(ch).n-1].qbeg + (ch).seeds[(ch).n-1].len)

int orc_mem_chain_flt(const orc_opt_t *opt, int n_chn, orc_chain_t *a)
{	/* upstream mem_chain_flt */
	int i, k;
	struct { size_t n, m; int *a; } chains = {0,0,0};
	if (n_chn == 0) return 0;
	for (i = k = 0; i < n_chn; ++i) {
		orc_chain_t *c = &a[i];
		c->first = -1; c->kept = 0;
		c->w = chain_weight(c);
		if ((int)c->w < opt->min_chain_weight) free(c->seeds);
		else a[k++] = *c;
	}
	n_chn = k;
	orc_introsort(a, n_chn, sizeof(orc_chain_t), flt_lt);
	a[0].kept = 3;
	PUSH(chains, int, 0);
	for (i = 1; i < n_chn; ++i) {
		int large_ovlp = 0;
		for (k = 0; k < (int)chains.n; ++k) {
			int j = chains.a[k];
			int b_max = chn_beg(a[j]) > chn_beg(a[i]) ? chn_beg(a[j]) : chn_beg(a[i]);
			int e_min = chn_end(a[j]) < chn_end(a[i]) ? chn_end(a[j]) : chn_end(a[i]);
			if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
				int li = chn_end(a[i]) - chn_beg(a[i]);
				int lj = chn_end(a[j]) - chn_beg(a[j]);
				int min_l = li < lj ? li : lj;
				if (e_min - b_max >= min_l * opt->mask_level && min_l < opt->max_chain_gap) {
					large_ovlp = 1;
					if (a[j].first < 0) a[j].first = i;
					if (a[i].w < a[j].w * opt->drop_ratio && (int)a[j].w - (int)a[i].w >= opt->min_seed_len << 1) break;
				}
			}
		}
		if (k == (int)chains.n) { PUSH(chains, int, i); a[i].kept = large_ovlp ? 2 : 3; }
	}
	for (i = 0; i < (int)chains.n; ++i) {
		orc_chain_t *c = &a[chains.a[i]];
		if (c->first >= 0) a[c->first].kept = 1;
	}
	free(chains.a);
	for (i = k = 0; i < n_chn; ++i) {
		if (a[i].kept == 0 || a[i].kept == 3) continue;
		if (++k >= opt->max_chain_extend) break;
	}
	for (; i < n_chn; ++i) if (a[i].kept < 3) a[i].kept = 0;
	for (i = k = 0; i < n_chn; ++i) {
		orc_chain_t *c = &a[i];
		if (c->kept == 0) free(c->seeds);
		else a[k++] = a[i];
	}
	return k;
}

/* ---------------- extension ---------------- */
static inline int cal_max_gap(const orc_opt_t *opt, int qlen)
{
	int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
	int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < opt->w << 1 ? l : opt->w << 1;
}

/* upstream bns_fetch_seq: clip [beg,end) to the contig (strand) holding mid and fetch bases */
static uint8_t *fetch_seq(const orc_idx_t *idx, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
	const orc_bns_t *bns = idx->bns;
	int64_t far_beg, far_end; int is_rev;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	assert(*beg <= mid && mid < *end);
	*rid = orc_bns_pos2rid(bns, orc_bns_depos(bns, mid, &is_rev));
	far_beg = bns->anns[*rid].offset;
	far_end = far_beg + bns->anns[*rid].len;
	if (is_rev) { int64_t t = far_beg; far_beg = (bns->l_pac << 1) - far_end; far_end = (bns->l_pac << 1) - t; }
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	uint8_t *seq = malloc(*end - *beg + 1);
	for (int64_t k = *beg; k < *end; ++k) seq[k - *beg] = orc_ref_base(idx->pac, bns->l_pac, k);
	return seq;
}

#define MAX_BAND_TRY 2

void orc_mem_chain2aln(const orc_opt_t *opt, const orc_idx_t *idx, int l_query, const uint8_t *query, const orc_chain_t *c, orc_alnreg_v *av)
{	/* upstream mem_chain2aln */
	int i, k, rid, max_off[2], aw[2];
	int64_t l_pac = idx->bns->l_pac, rmax[2], tmp, max = 0;
	const orc_seed_t *s;
	uint8_t *rseq = 0;
	uint64_t *srt;
	if (c->n == 0) return;
	rmax[0] = l_pac << 1; rmax[1] = 0;
	for (i = 0; i < c->n; ++i) {
		int64_t b, e;
		const orc_seed_t *t = &c->seeds[i];
		b = t->rbeg - (t->qbeg + cal_max_gap(opt, t->qbeg));
		e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + cal_max_gap(opt, l_query - t->qbeg - t->len));
		rmax[0] = rmax[0] < b ? rmax[0] : b;
		rmax[1] = rmax[1] > e ? rmax[1] : e;
		if (t->len > max) max = t->len;
	}
	rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
	rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
	if (rmax[0] < l_pac && l_pac < rmax[1]) {
		if (c->seeds[0].rbeg < l_pac) rmax[1] = l_pac;
		else rmax[0] = l_pac;
	}
	rseq = fetch_seq(idx, &rmax[0], c->seeds[0].rbeg, &rmax[1], &rid);
	assert(c->rid == rid);
	srt = malloc(c->n * 8);
	for (i = 0; i < c->n; ++i) srt[i] = (uint64_t)c->seeds[i].score << 32 | i;
	orc_introsort_u64(c->n, srt);
	for (k = c->n - 1; k >= 0; --k) {
		orc_alnreg_t *a;
		s = &c->seeds[(uint32_t)srt[k]];
		for (i = 0; i < (int)av->n; ++i) { /* already covered by an earlier extension? */
			orc_alnreg_t *p = &av->a[i];
			int64_t rd; int qd, w, max_gap;
			if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;
			if (s->len - p->seedlen0 > .1 * l_query) continue;
			qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
			max_gap = cal_max_gap(opt, qd < rd ? qd : rd);
			w = max_gap < p->w ? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
			qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
			max_gap = cal_max_gap(opt, qd < rd ? qd : rd);
			w = max_gap < p->w ? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
		}
		if (i < (int)av->n) {
			for (i = k + 1; i < c->n; ++i) { /* overlapping seeds that disagree on the diagonal */
				const orc_seed_t *t;
				if (srt[i] == 0) continue;
				t = &c->seeds[(uint32_t)srt[i]];
				if (t->len < s->len * .95) continue;
				if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
				if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
			}
			if (i == c->n) { srt[k] = 0; continue; }
		}
		if (av->n == av->m) { av->m = av->m ? av->m << 1 : 4; av->a = realloc(av->a, av->m * sizeof(orc_alnreg_t)); }
		a = &av->a[av->n++];
		memset(a, 0, sizeof(orc_alnreg_t));
		a->w = aw[0] = aw[1] = opt->w;
		a->score = a->truesc = -1;
		a->rid = c->rid;
		if (s->qbeg) { /* left extension */
			uint8_t *rs, *qs; int qle, tle, gtle, gscore;
			qs = malloc(s->qbeg);
			for (i = 0; i < s->qbeg; ++i) qs[i] = query[s->qbeg - 1 - i];
			tmp = s->rbeg - rmax[0];
			rs = malloc(tmp + 1);
			for (i = 0; i < tmp; ++i) rs[i] = rseq[tmp - 1 - i];
			for (i = 0; i < MAX_BAND_TRY; ++i) {
				int prev = a->score;
				aw[0] = opt->w << i;
				a->score = orc_ksw_extend2(s->qbeg, qs, (int)tmp, rs, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, aw[0], opt->pen_clip5, opt->zdrop, s->len * opt->a, &qle, &tle, &gtle, &gscore, &max_off[0]);
				if (a->score == prev || max_off[0] < (aw[0] >> 1) + (aw[0] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip5) { a->qb = s->qbeg - qle; a->rb = s->rbeg - tle; a->truesc = a->score; }
			else { a->qb = 0; a->rb = s->rbeg - gtle; a->truesc = gscore; }
			free(qs); free(rs);
		} else a->score = a->truesc = s->len * opt->a, a->qb = 0, a->rb = s->rbeg;
		if (s->qbeg + s->len != l_query) { /* right extension */
			int qle, tle, qe, re, gtle, gscore, sc0 = a->score;
			qe = s->qbeg + s->len;
			re = (int)(s->rbeg + s->len - rmax[0]);
			assert(re >= 0);
			for (i = 0; i < MAX_BAND_TRY; ++i) {
				int prev = a->score;
				aw[1] = opt->w << i;
				a->score = orc_ksw_extend2(l_query - qe, query + qe, (int)(rmax[1] - rmax[0] - re), rseq + re, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, aw[1], opt->pen_clip3, opt->zdrop, sc0, &qle, &tle, &gtle, &gscore, &max_off[1]);
				if (a->score == prev || max_off[1] < (aw[1] >> 1) + (aw[1] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip3) { a->qe = qe + qle; a->re = rmax[0] + re + tle; a->truesc += a->score - sc0; }
			else { a->qe = l_query; a->re = rmax[0] + re + gtle; a->truesc += gscore - sc0; }
		} else a->qe = l_query, a->re = s->rbeg + s->len;
		for (i = 0, a->seedcov = 0; i < c->n; ++i) {
			const orc_seed_t *t = &c->seeds[i];
			if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re) a->seedcov += t->len;
		}
		a->w = aw[0] > aw[1] ? aw[0] : aw[1];
		a->seedlen0 = s->len;
		a->frac_rep = c->frac_rep;
	}
	free(srt); free(rseq);
}

/* ---------------- region sort / dedup / patch ---------------- */
#define PATCH_MAX_R_BW 0.05f
#define PATCH_MIN_SC_RATIO 0.90f

static int patch_reg(const orc_opt_t *opt, const orc_idx_t *idx, uint8_t *query, const orc_alnreg_t *a, const orc_alnreg_t *b, int *_w)
{	/* upstream mem_patch_reg */
	int w, score, q_s, r_s; double r;
	if (idx == 0 || query == 0) return 0;
	assert(a->rid == b->rid && a->rb <= b->rb);
	if (a->rb < idx->bns->l_pac && b->rb >= idx->bns->l_pac) return 0;
	if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
	w = (int)((a->re - b->rb) - (a->qe - b->qb));
	w = w > 0 ? w : -w;
	r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
	r = r > 0. ? r : -r;
	if (a->re < b->rb || a->qe < b->qb) {
		if (w > opt->w << 1 || r >= PATCH_MAX_R_BW) return 0;
	} else if (w > opt->w << 2 || r >= PATCH_MAX_R_BW * 2) return 0;
	w += a->w + b->w;
	w = w < opt->w << 2 ? w : opt->w << 2;
	orc_gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, idx->bns->l_pac, idx->pac, b->qe - a->qb, query + a->qb, a->rb, b->re, &score, 0, 0);
	q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
	r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) return 0;
	*_w = w;
	return score;
}

static int ars2_lt(const void *a, const void *b) { return ((const orc_alnreg_t*)a)->re < ((const orc_alnreg_t*)b)->re; }
static int ars_lt(const void *a_, const void *b_)
{
	const orc_alnreg_t *a = a_, *b = b_;
	return a->score > b->score || (a->score == b->score && (a->rb < b->rb || (a->rb == b->rb && a->qb < b->qb)));
}

int orc_mem_sort_dedup_patch(const orc_opt_t *opt, const orc_idx_t *idx, uint8_t *query, int n, orc_alnreg_t *a)
{	/* upstream mem_sort_dedup_patch */
	int m, i, j;
	if (n <= 1) return n;
	orc_introsort(a, n, sizeof(orc_alnreg_t), ars2_lt);
	for (i = 0; i < n; ++i) a[i].n_comp = 1;
	for (i = 1; i < n; ++i) {
		orc_alnreg_t *p = &a[i];
		if (p->rid != a[i-1].rid || p->rb >= a[i-1].re + opt->max_chain_gap) continue;
		for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
			orc_alnreg_t *q = &a[j];
			int64_t or_, oq, mr, mq; int score, w;
			if (q->qe == q->qb) continue;
			or_ = q->re - p->rb;
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			} else if (q->rb < p->rb && (score = patch_reg(opt, idx, query, q, p, &w)) > 0) {
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub ? p->sub : q->sub;
				p->csub = p->csub > q->csub ? p->csub : q->csub;
				p->qb = q->qb; p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
	n = m;
	orc_introsort(a, n, sizeof(orc_alnreg_t), ars_lt);
	for (i = 1; i < n; ++i)
		if (a[i].score == a[i-1].score && a[i].rb == a[i-1].rb && a[i].qb == a[i-1].qb) a[i].qe = a[i].qb;
	for (i = 1, m = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
	return m;
}

#define MEM_MINSC_COEF 5.5f
#define MEM_SEEDSW_COEF 0.05f

orc_alnreg_v orc_mem_align1_core(const orc_opt_t *opt, const orc_idx_t *idx, int l_seq, uint8_t *seq)
{	/* upstream mem_align1_core; seq already nt4-coded (0..4) */
	orc_alnreg_v regs = {0,0,0};
	orc_chain_v chn = orc_mem_chain(opt, idx, l_seq, seq);
	chn.n = orc_mem_chain_flt(opt, (int)chn.n, chn.a);
	{	/* upstream mem_flt_chained_seeds: only active when MEM_MINSC_COEF*log(l) <= MEM_SEEDSW_COEF*l (l >~ 1 kb) */
		double min_l = opt->min_chain_weight ? 2.8f * opt->min_chain_weight : MEM_MINSC_COEF * log(l_seq);
		assert(min_l > MEM_SEEDSW_COEF * l_seq && "reads >= ~1 kb are outside the oracle's scope");
	}
	for (size_t i = 0; i < chn.n; ++i) {
		orc_mem_chain2aln(opt, idx, l_seq, seq, &chn.a[i], &regs);
		free(chn.a[i].seeds);
	}
	free(chn.a);
	regs.n = orc_mem_sort_dedup_patch(opt, idx, seq, (int)regs.n, regs.a);
	return regs;
}
