/*
 * k_extlane.h -- seed extension with ONE LANE PER EXTENSION (upstream
 * mem_chain2aln's calls of ksw_extend2), for the seed every chain extends first.
 *
 * mem_chain2aln walks the seeds of a chain from the best one down and skips a seed that is already
 * contained in an earlier region of the read, so within a read the *decision* to extend is sequential.
 * The *result* of an extension, however, depends only on (read, chain window, seed): it can be computed
 * ahead of the decision.  The first (best) seed of a chain is extended in all but a few cases, and
 * for repeat-heavy reads (hundreds of chains, one extended seed each) these extensions are ~all of the
 * Smith-Waterman work.  So:
 *   ssg_k_ext_prep   one lane per chain: reference window (rmax) and best seed -> job record
 *   ssg_k_ext_lane   one lane per job and side (left, then right with the left score as h0): a scalar
 *                    restatement of ksw_extend2 per lane, 64 independent extensions per wavefront.  The
 *                    DP row {H,E} and the query live in LDS, one 32-bit word per column and lane
 *                    (h:16 | e:13 | query code:3; word (j*64 + lane) -> bank = lane, conflict free for
 *                    any mix of per-lane columns).  Jobs are sorted by query-side length so the lanes of
 *                    a wave run similar trip counts.  The reference is read straight from the 2-bit pac
 *                    (one aligned 32-bit word = 16 bases, next word prefetched).
 *   ssg_k_chain2aln  (k_extend.h) then replays upstream's decisions per read and consumes the results;
 *                    the rare later seed of a chain that is extended too still goes through the
 *                    wave-per-extension kernel code there.
 * No cross-lane operation is used here: per-lane control flow is plain SIMT divergence.
 */
#ifndef SSG_K_EXTLANE_H
#define SSG_K_EXTLANE_H
#include "k_sw.h"

/* everything ssg_k_chain2aln needs to know about a chain, in one record at a computable address (chain_off[r] + position) */
struct ssg_xjob_t { int64_t rbeg, rmax0, rmax1; int32_t read, flag; int16_t qbeg, len, l_query, seed_t; int32_t cn, rid, first_seed; float frac_rep; };   /* 56 bytes */
struct ssg_xres_t { int32_t score, qle, tle, gtle, gscore, max_off, aw, done; };                            /* 32 bytes */

SSG_DEVFN int ssg_cal_max_gap2(const ssg_mem_opt_t &opt, int qlen)
{	/* upstream cal_max_gap */
	int l_del = (int)((double)(qlen * opt.a - opt.o_del) / opt.e_del + 1.);
	int l_ins = (int)((double)(qlen * opt.a - opt.o_ins) / opt.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < opt.w << 1 ? l : opt.w << 1;
}

/* reference bases along an extension: doubled coordinate p0 + i*dir, read from the forward-strand pac */
struct ssg_tgt_t {
	const uint8_t *pac; int64_t f, nbytes, cur_i; int fs, comp; uint32_t cur, nxt;
};
SSG_DEVFN uint32_t ssg_pac_word(const ssg_tgt_t &t, int64_t wi)
{	/* aligned 32-bit word wi of the pac (16 bases), clamped to the array: (l_pac/4 + 1) bytes */
	const int64_t last = (t.nbytes - 1) >> 2;
	wi = wi < 0 ? 0 : wi > last ? last : wi;
	const int64_t b = wi << 2;
	if (b + 4 <= t.nbytes) return ((const uint32_t*)t.pac)[wi];
	uint32_t v = 0;
	for (int k = 0; k < 4; ++k) if (b + k < t.nbytes) v |= (uint32_t)t.pac[b + k] << (8 * k);
	return v;
}
SSG_DEVFN void ssg_tgt_init(ssg_tgt_t &t, const ssg_index_view_t &ix, int64_t p0, int dir)
{
	const int64_t l_pac = ix.l_pac;
	t.pac = ix.pac; t.nbytes = (l_pac >> 2) + 1;
	if (p0 < l_pac) { t.f = p0; t.fs = dir; t.comp = 0; } else { t.f = (l_pac << 1) - 1 - p0; t.fs = -dir; t.comp = 1; }
	t.cur_i = t.f >> 4;
	t.cur = ssg_pac_word(t, t.cur_i); t.nxt = ssg_pac_word(t, t.cur_i + t.fs);
}
SSG_DEVFN int ssg_tgt_next(ssg_tgt_t &t)
{	/* base at t.f, then advance; pac byte k>>2 holds base k at bits (~k&3)*2 and a 32-bit load is little endian */
	const int64_t wi = t.f >> 4;
	if (wi != t.cur_i) { t.cur = t.nxt; t.cur_i = wi; t.nxt = ssg_pac_word(t, wi + t.fs); }
	const int k = (int)(t.f & 15);
	const int base = (int)(t.cur >> (((k >> 2) << 3) + ((~k & 3) << 1))) & 3;
	t.f += t.fs;
	return t.comp ? 3 - base : base;
}

/* How many rows an extension is likely to run, from its ungapped diagonal alone: ksw_extend2 leaves its row loop when the row maximum falls zdrop below the
 * best one (or to zero); without that it runs all tlen rows.  The lanes of a wave run their rows in step, so a wave lasts as long as its longest lane: jobs are
 * ordered by side length AND by this figure (measured on the bench batch before: the lanes of the long classes ran 78 rows on average in waves of 163).  A sort
 * key only -- no result depends on it. */
SSG_DEVFN int ssg_xl_pred_rows(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const uint8_t *q, int qstep, int qlen, int tlen, int64_t p0, int dir, int h0)
{
	if (qlen <= 0 || tlen <= 0) return 0;
	ssg_tgt_t tg;
	ssg_tgt_init(tg, ix, p0, dir);
	const int n = qlen < tlen ? qlen : tlen;
	int sc = h0, mx = h0;
	for (int i = 0; i < n; ++i, q += qstep) {
		const int tb = ssg_tgt_next(tg), qb = (int)*q;
		sc += qb > 3 ? -1 : qb == tb ? opt.a : -opt.b;
		if (sc > mx) mx = sc;
		else if (sc <= 0 || (opt.zdrop > 0 && mx - sc > opt.zdrop)) return i + 1;
	}
	return tlen;
}

/* one lane per surviving chain g (global numbering: chain_off[r] + position in the read's order[]) */
__global__ void __launch_bounds__(256) ssg_k_ext_prep(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, long n_jobs, const int64_t *read_off,
                               const int64_t *seed_off, const ssg_seed_t *seeds, const ssg_chain_t *chains, const int32_t *order,
                               const int32_t *chain_seeds, const int64_t *chain_off, int twin_cap,
                               ssg_xjob_t *jobs, uint64_t *key_l, uint64_t *key_r, int short_cap, unsigned int *n_long /* [2]: sides longer than short_cap */,
                               const uint8_t *seq, int rows_key /* 0: order by side length alone */)
{
	const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_jobs) return;
	int lo = 0, hi = n_reads - 1;   /* last read with chain_off[r] <= g */
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (chain_off[mid] <= g) lo = mid; else hi = mid - 1; }
	const int r = lo, ci = (int)(g - chain_off[r]);
	const long s0 = seed_off[r];
	const int l_query = (int)(read_off[r+1] - read_off[r]);
	const ssg_chain_t c = chains[s0 + order[s0 + ci]];
	const int32_t *cs = chain_seeds + c.first_seed;
	const int64_t l_pac = ix.l_pac;
	ssg_xjob_t jb;
	jb.read = r; jb.flag = 0; jb.l_query = (int16_t)l_query; jb.rbeg = 0; jb.rmax0 = jb.rmax1 = 0; jb.qbeg = jb.len = 0; jb.seed_t = 0;
	jb.cn = c.n; jb.rid = c.rid; jb.first_seed = c.first_seed; jb.frac_rep = c.frac_rep;
	if (c.n == 0) { jb.flag = 2; jobs[g] = jb; key_l[g] = (uint64_t)511 << 41 | (uint64_t)g; key_r[g] = (uint64_t)511 << 41 | (uint64_t)g; return; }
	int64_t rmax0 = l_pac << 1, rmax1 = 0;
	uint64_t best = 0; int best_t = 0;
	for (int i = 0; i < c.n; ++i) {
		const ssg_seed_t t = seeds[cs[i]];
		const int64_t b = t.rbeg - (t.qbeg + ssg_cal_max_gap2(opt, t.qbeg));
		const int64_t e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + ssg_cal_max_gap2(opt, l_query - t.qbeg - t.len));
		rmax0 = rmax0 < b ? rmax0 : b;
		rmax1 = rmax1 > e ? rmax1 : e;
		const uint64_t k = (uint64_t)t.score << 32 | (uint64_t)i;   /* upstream sorts (score<<32 | i) and starts from the largest */
		if (i == 0 || k > best) { best = k; best_t = i; }
	}
	rmax0 = rmax0 > 0 ? rmax0 : 0;
	rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
	const int64_t rbeg0 = seeds[cs[0]].rbeg;
	if (rmax0 < l_pac && l_pac < rmax1) { if (rbeg0 < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
	{	/* upstream bns_fetch_seq: clip to the contig holding the first seed */
		int is_rev; const int rid = ssg_pos2rid(ix, ssg_depos(ix, rbeg0, &is_rev));
		int64_t far_beg = ix.ctg_off[rid], far_end = far_beg + ix.ctg_len[rid];
		if (is_rev) { const int64_t t2 = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t2; }
		rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
		rmax1 = rmax1 < far_end ? rmax1 : far_end;
	}
	const ssg_seed_t s = seeds[cs[best_t]];
	jb.rbeg = s.rbeg; jb.rmax0 = rmax0; jb.rmax1 = rmax1; jb.qbeg = (int16_t)s.qbeg; jb.len = (int16_t)s.len; jb.seed_t = (int16_t)best_t;
	if (rmax1 - rmax0 > twin_cap) jb.flag = 1;   /* window beyond the wave kernel's buffer: reported as an error there */
	jobs[g] = jb;
	const int ql = jb.flag ? 0 : s.qbeg, qr = jb.flag ? 0 : l_query - s.qbeg - s.len;
	int pl = 0, pr = 0;
	if (rows_key && !jb.flag) {
		const uint8_t *query = seq + read_off[r];
		pl = ssg_xl_pred_rows(ix, opt, query + s.qbeg - 1, -1, ql, (int)(s.rbeg - rmax0), s.rbeg - 1, -1, s.len * opt.a);
		pr = ssg_xl_pred_rows(ix, opt, query + s.qbeg + s.len, 1, qr, (int)(rmax1 - (s.rbeg + s.len)), s.rbeg + s.len, 1, s.len * opt.a);
		pl = pl < 511 ? pl : 511; pr = pr < 511 ? pr : 511;
	}
	/* ascending sort = longest query side first (511 = nothing to do on this side), and among sides of one length the one expected to run most rows first */
	key_l[g] = (uint64_t)(511 - ql) << 41 | (uint64_t)(511 - pl) << 32 | (uint64_t)g;
	key_r[g] = (uint64_t)(511 - qr) << 41 | (uint64_t)(511 - pr) << 32 | (uint64_t)g;
	{	/* one atomic per wave and side */
		const unsigned long long bl = wv_ballot(ql > short_cap), br = wv_ballot(qr > short_cap), act = wv_ballot(1);
		if (wv_lane() == (int)__builtin_ctzll(act)) { if (bl) atomicAdd(&n_long[0], (unsigned)__popcll(bl)); if (br) atomicAdd(&n_long[1], (unsigned)__popcll(br)); }
	}
}

/* LDS word of a column: h:13 | e:13 | 6 x query code:6 (DP values stay below 8191: checked on the host).  The query field is the
 * bit offset of the column's score in the row's score table T (five signed 6-bit fields, rebuilt per target base). */
#define SSG_XL_QS(wd) ((unsigned)((wd) >> 26))
#define SSG_XL_E(wd)  ((int)(((wd) >> 13) & 0x1fff))
#define SSG_XL_H(wd)  ((int)((wd) & 0x1fff))
#define SSG_XL_QMASK  0xfc000000u
#define SSG_XL_QWORD(code) ((uint32_t)((code) * 6) << 26)
#ifdef SSG_EMU
SSG_DEVFN int ssg_sbfe6(uint32_t t, unsigned off) { return (int)(t << (26 - off)) >> 26; }
#else
SSG_DEVFN int ssg_sbfe6(uint32_t t, unsigned off) { return __builtin_amdgcn_sbfe((int)t, off, 6u); }
#endif

/* upstream ksw_extend2, one lane; Lc[j*64] is this lane's column j (query field already set); U = columns per trip of the cell loop */
template <int U>
SSG_DEVFN ssg_ext_res_t ln_extend2(const ssg_mem_opt_t &opt, const ssg_index_view_t &ix, uint32_t *Lc, int qlen, int tlen, int64_t p0, int dir,
                                   int w, int end_bonus, int zdrop, int h0, unsigned long long *cells, unsigned int *tune_rowmax = 0, unsigned long long *tune_lane = 0)
{
	const int sa = opt.a, sb = opt.b;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int i, j, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
	{	/* first row */
		int prev = h0;
		for (j = 0; j <= qlen; ++j) {
			int h;
			if (j == 0) h = h0;
			else if (j == 1) h = h0 > oe_ins ? h0 - oe_ins : 0;
			else h = prev > e_ins ? prev - e_ins : 0;
			prev = h;
			Lc[j * 64] = (Lc[j * 64] & SSG_XL_QMASK) | (uint32_t)h;
		}
	}
	{	/* band clamp */
		const int mx = sa > 0 ? sa : 0;
		max_ins = (int)((double)(qlen * mx + end_bonus - o_ins) / e_ins + 1.);
		max_ins = max_ins > 1 ? max_ins : 1;
		w = w < max_ins ? w : max_ins;
		max_del = (int)((double)(qlen * mx + end_bonus - o_del) / e_del + 1.);
		max_del = max_del > 1 ? max_del : 1;
		w = w < max_del ? w : max_del;
	}
	max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
	beg = 0; end = qlen;
	unsigned long long ncell = 0;
	ssg_tgt_t tg;
	ssg_tgt_init(tg, ix, p0, dir);
	/* score table of a row: field q (bits 6q..6q+5) = score of query code q against the row's target base */
	uint32_t t_mis = (uint32_t)(-1 & 63) << 24;
	for (int q = 0; q < 4; ++q) t_mis |= (uint32_t)(-sb & 63) << (6 * q);
	for (i = 0; i < tlen; ++i) {
		int f = 0, h1, mm = 0, mj = -1;
		const int tb = ssg_tgt_next(tg);
		const uint32_t T = (t_mis & ~(63u << (6 * tb))) | (uint32_t)(sa & 63) << (6 * tb);
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
		else h1 = 0;
		if (SSG_TUNING && tune_rowmax) {   /* rows of a wave's lanes run in step: the wave pays the widest lane's trips in every row */
			const unsigned trips = end > beg ? (unsigned)((end - beg + U - 1) / U) : 0u;
			atomicMax(&tune_rowmax[i < 511 ? i : 511], trips + 1u);
			tune_lane[0] += trips; tune_lane[1] += 1;
		}
		if (end > beg) {
			ncell += (unsigned long long)(end - beg);
			int mk = -1;                                 /* max over the row of (h << 9 | j): the largest h, at its last column (columns up to 320) */
			auto cell = [&](const uint32_t wd, const int jj) -> uint32_t {
				int M = SSG_XL_H(wd), e = SSG_XL_E(wd), h, t;
				const int sc = ssg_sbfe6(T, SSG_XL_QS(wd));
				M = M ? M + sc : 0;
				h = M > e ? M : e;
				h = h > f ? h : f;
				{ const int k = h << 9 | jj; mk = mk > k ? mk : k; }
				t = M - oe_del; t = t > 0 ? t : 0;
				e -= e_del; e = e > t ? e : t;
				const uint32_t out = (wd & SSG_XL_QMASK) | ((uint32_t)e << 13) | (uint32_t)h1;
				h1 = h;
				t = M - oe_ins; t = t > 0 ? t : 0;
				f -= e_ins; f = f > t ? f : t;
				return out;
			};
			/* U columns per trip, the next U in flight meanwhile (the long class runs one wave per SIMD: nothing else hides LDS latency).  Whole trips carry no
			 * per-column test (five instructions of twenty-six a cell went into `is this column still in the band'); the last columns of the row follow once */
			uint32_t wc[U];
			SSG_UNROLL for (int u = 0; u < U; ++u) wc[u] = Lc[(beg + u) * 64];
			for (j = beg; j + U <= end; j += U) {
				uint32_t wn[U];
				SSG_UNROLL for (int u = 0; u < U; ++u) wn[u] = Lc[(j + U + u) * 64];
				SSG_UNROLL for (int u = 0; u < U; ++u) Lc[(j + u) * 64] = cell(wc[u], j + u);
				SSG_UNROLL for (int u = 0; u < U; ++u) wc[u] = wn[u];
			}
			SSG_UNROLL for (int u = 0; u < U - 1; ++u) if (j + u < end) Lc[(j + u) * 64] = cell(wc[u], j + u);
			mm = mk >> 9; mj = mk & 511;                 /* end > beg: at least one column */
			j = end;
		} else j = beg;
		Lc[end * 64] = (Lc[end * 64] & SSG_XL_QMASK) | (uint32_t)h1;
		if (j == qlen) {
			max_ie = gscore > h1 ? max_ie : i;
			gscore = gscore > h1 ? gscore : h1;
		}
		if (mm == 0) break;
		if (mm > max) {
			max = mm; max_i = i; max_j = mj;
			max_off = max_off > iabs(mj - i) ? max_off : iabs(mj - i);
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) { if (max - mm - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
			else { if (max - mm - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
		}
		for (j = beg; j < end && (Lc[j * 64] & ~SSG_XL_QMASK) == 0; ++j);
		beg = j;
		for (j = end; j >= beg && (Lc[j * 64] & ~SSG_XL_QMASK) == 0; --j);
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	if (cells) *cells += ncell;
	ssg_ext_res_t r; r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

#define SSG_XL_BAND_TRY 2   /* == SSG_MAX_BAND_TRY (upstream MAX_BAND_TRY) */

/* side 0: left extensions (query and reference walked backwards from the seed); side 1: right extensions.
 * sorted[t] = (511 - side length) << 41 | (511 - expected rows) << 32 | job id; QCAP+1 columns of LDS per lane. */
/* one job of one side; L: the workgroup's (qcap + 2 U) x 64 words of LDS */
template <int U>
SSG_DEVFN void ssg_ext_lane_job(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const int side, const long job_first, const long n_jobs, const uint64_t *sorted,
                                const ssg_xjob_t *jobs, const uint8_t *seq, const int64_t *read_off, ssg_xres_t *res_l, ssg_xres_t *res_r,
                                unsigned long long *cells, uint32_t *L, const int qcap)
{
	const long t = job_first + (long)blockIdx.x * 64 + threadIdx.x;
#ifdef SSG_TUNE   /* lane utilisation of this wave: per row index the widest lane's trips (rows run in step), against the lanes' own sums; slots 32.. (U = 2) / 40.. (U = 4) */
	__shared__ unsigned int tune_rowmax[SSG_XL_BAND_TRY][512];
	for (int k = (int)threadIdx.x; k < SSG_XL_BAND_TRY * 512; k += 64) (&tune_rowmax[0][0])[k] = 0;
	unsigned long long tune_lane[2] = { 0, 0 };
	__syncthreads();
#define SSG_XL_TUNE_ARGS(i) , tune_rowmax[i], tune_lane
#else
#define SSG_XL_TUNE_ARGS(i)
#endif
	if (t >= n_jobs) return;
	const uint64_t key = sorted[t];
	if ((key >> 41) >= 511) return;   /* nothing on this side */
	const long g = (long)(uint32_t)key;
	const ssg_xjob_t jb = jobs[g];
	if (jb.flag) return;
	uint32_t *Lc = L + (threadIdx.x & 63);
	const uint8_t *query = seq + read_off[jb.read];
	unsigned long long nc = 0;
	ssg_ext_res_t x; x.score = -1; x.qle = x.tle = x.gtle = x.gscore = x.max_off = 0;
	int aw = opt.w, score = -1;
	if (side == 0) {
		const int qlen = jb.qbeg;
		if (qlen <= 0 || qlen > qcap) return;
		for (int j = 0; j < qlen; ++j) Lc[j * 64] = SSG_XL_QWORD(query[jb.qbeg - 1 - j]);
		Lc[qlen * 64] = 0;
		const int tlen = (int)(jb.rbeg - jb.rmax0);
		for (int i = 0; i < SSG_XL_BAND_TRY; ++i) {
			const int prev = score;
			aw = opt.w << i;
			x = ln_extend2<U>(opt, ix, Lc, qlen, tlen, jb.rbeg - 1, -1, aw, opt.pen_clip5, opt.zdrop, jb.len * opt.a, &nc SSG_XL_TUNE_ARGS(i));
			score = x.score;
			if (score == prev || x.max_off < (aw >> 1) + (aw >> 2)) break;
		}
		ssg_xres_t o; o.score = x.score; o.qle = x.qle; o.tle = x.tle; o.gtle = x.gtle; o.gscore = x.gscore; o.max_off = x.max_off; o.aw = aw; o.done = 1;
		res_l[g] = o;
	} else {
		const int qe = jb.qbeg + jb.len, qlen = jb.l_query - qe;
		if (qlen <= 0 || qlen > qcap) return;
		const int sc0 = jb.qbeg ? res_l[g].score : jb.len * opt.a;
		for (int j = 0; j < qlen; ++j) Lc[j * 64] = SSG_XL_QWORD(query[qe + j]);
		Lc[qlen * 64] = 0;
		const int tlen = (int)(jb.rmax1 - (jb.rbeg + jb.len));
		score = sc0;
		for (int i = 0; i < SSG_XL_BAND_TRY; ++i) {
			const int prev = score;
			aw = opt.w << i;
			x = ln_extend2<U>(opt, ix, Lc, qlen, tlen, jb.rbeg + jb.len, 1, aw, opt.pen_clip3, opt.zdrop, sc0, &nc SSG_XL_TUNE_ARGS(i));
			score = x.score;
			if (score == prev || x.max_off < (aw >> 1) + (aw >> 2)) break;
		}
		ssg_xres_t o; o.score = x.score; o.qle = x.qle; o.tle = x.tle; o.gtle = x.gtle; o.gscore = x.gscore; o.max_off = x.max_off; o.aw = aw; o.done = 1;
		res_r[g] = o;
	}
	if (cells && nc) atomicAdd(cells, nc);
#ifdef SSG_TUNE
	{
		const int sl = U == 2 ? 32 : 40;
		atomicAdd(&ssg_dbg_cyc[sl], tune_lane[0]); atomicAdd(&ssg_dbg_cyc[sl + 1], tune_lane[1]); atomicAdd(&ssg_dbg_cyc[sl + 2], 1ull);   /* lanes' trips, lanes' rows, lanes that had a side */
		if (wv_lane() == (int)__builtin_ctzll(wv_ballot(1))) {   /* the wave's: sum over rows of the widest lane's trips, and its rows */
			unsigned long long wt = 0, wr = 0;
			for (int k = 0; k < SSG_XL_BAND_TRY * 512; ++k) { const unsigned v = (&tune_rowmax[0][0])[k]; if (v) { wt += v - 1; ++wr; } }
			atomicAdd(&ssg_dbg_cyc[sl + 3], wt); atomicAdd(&ssg_dbg_cyc[sl + 4], wr); atomicAdd(&ssg_dbg_cyc[sl + 5], 1ull);
		}
	}
#endif
}

template <int QCAP>
__global__ void __launch_bounds__(64) ssg_k_ext_lane(ssg_index_view_t ix, ssg_mem_opt_t opt, int side, long job_first, long n_jobs, const uint64_t *sorted,
                               const ssg_xjob_t *jobs, const uint8_t *seq, const int64_t *read_off, ssg_xres_t *res_l, ssg_xres_t *res_r,
                               unsigned long long *cells)
{
	constexpr int U = QCAP > 72 ? 4 : 2;
	__shared__ uint32_t L[(QCAP + 2 * U) * 64];   /* columns 0..qlen, and the 2U - 1 the cell loop may read ahead */
	ssg_ext_lane_job<U>(ix, opt, side, job_first, n_jobs, sorted, jobs, seq, read_off, res_l, res_r, cells, L, QCAP);
}

/* The same with as much LDS as the launch asks for ((qcap + 2 U) x 256 bytes): the host cuts the sorted job list into classes of 8 more columns each, so that a wave
 * holds what its longest side needs -- 73..80 columns: 22 KB, seven waves a CU; 129..136: 37 KB, four (what the fixed class above gives every side beyond 72); a side
 * of 100 of a 250-base read: 28 KB instead of the 68 KB of the 256-column class (two waves a CU). */
template <int U>
__global__ void __launch_bounds__(64) ssg_k_ext_lane_dyn(ssg_index_view_t ix, ssg_mem_opt_t opt, int side, long job_first, long n_jobs, const uint64_t *sorted,
                               const ssg_xjob_t *jobs, const uint8_t *seq, const int64_t *read_off, ssg_xres_t *res_l, ssg_xres_t *res_r,
                               unsigned long long *cells, int qcap)
{
#ifdef SSG_EMU
	uint32_t *L = (uint32_t*)emu::dyn_lds;
#else
	extern __shared__ uint32_t ssg_xl_lds[];
	uint32_t *L = ssg_xl_lds;
#endif
	ssg_ext_lane_job<U>(ix, opt, side, job_first, n_jobs, sorted, jobs, seq, read_off, res_l, res_r, cells, L, qcap);
}

/* out[j] = number of keys whose side-length field (bits 41..49) is below t[j], in a list sorted ascending by it (= sides longer than 511 - t[j]): a binary search per threshold */
struct ssg_thr64_t { int n; int t[63]; };
__global__ void ssg_k_sorted_hi_below(const uint64_t *sorted, long n, ssg_thr64_t th, unsigned int *out)
{
	const int j = (int)threadIdx.x;
	if (j >= th.n) return;
	const uint64_t t = (uint64_t)th.t[j];
	long lo = 0, hi = n;
	while (lo < hi) { const long mid = (lo + hi) >> 1; if ((sorted[mid] >> 41) < t) lo = mid + 1; else hi = mid; }
	out[j] = (unsigned int)lo;
}
#endif
