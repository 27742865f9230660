/*
 * k_sw.h -- one-wavefront-per-job Smith-Waterman primitives for gfx950 (SURVEY.md 8a rows a7,
 * a10, a12): upstream ksw_extend2, ksw_global2 (+backtrace) and ksw_align2's contract.
 *
 * Layout: lane l of the wave owns query columns j = l + 64*s (s < NS <= 4, i.e. queries up to 255
 * bases, enough for 2x150 and 2x250); H(i-1,j-1)/E(i,j) of upstream's eh[] array live in VGPRs for
 * the whole job, exactly one register pair per column, so even upstream's reads of stale eh[]
 * entries beyond a shrunken band are reproduced for free.  A DP row is processed by all lanes at
 * once:
 *   - the diagonal operand H(i-1,j-1) is the lane's own register (eh[j].h);
 *   - H(i,j-1) for the next row comes from lane j-1 via a wave shift (DPP row_shr/wave_shr);
 *   - the horizontal gap F(i,j) depends only on the row's M values (upstream opens E and F from M,
 *     not from H), so F is a max-plus prefix scan: F(i,j) = max_k<j (max(M_k-oe,0) - (j-1-k)*e),
 *     evaluated with a 6-step wave scan; for the textbook local recurrence the same holds with
 *     M replaced by max(M,E,0) because an F-opened-from-F term is dominated when o_ins > 0;
 *   - row maximum / arg-max, band trimming and the z-drop test are wave reductions + ballots, so
 *     beg/end/max/... stay wave-uniform (SGPR-resident after readfirstlane).
 * All arithmetic is int32 like upstream: results are bit-exact.  No MFMA: this is integer DP.
 */
#ifndef SSG_K_SW_H
#define SSG_K_SW_H
#include "ssg_dev.h"

#define SSG_NEG (-(1 << 29))

/* sequence accessor: base k of a byte-coded sequence walked with stride +1/-1 */
struct ssg_seqv_t { const uint8_t *p; int dir; };
SSG_DEVFN int sq_at(const ssg_seqv_t &s, int k) { return s.p[s.dir * k]; }

/* value of column j-1's v for column j = lane + 64*s ; `first` is returned for j == 0 */
template <int NS>
SSG_DEVFN void wv_shift_cols(const int (&v)[NS], int (&out)[NS], int first)
{
	int lane = wv_lane(), carry = first;
	SSG_UNROLL for (int s = 0; s < NS; ++s) {
		int up = wv_shfl(v[s], lane - 1);
		int last = wv_shfl(v[s], 63);
		out[s] = lane == 0 ? carry : up;
		carry = last;
	}
}

/* ------------------------------------------------------------------------------------------
 * upstream ksw_extend2.  Returns the score in all lanes; *res filled (uniform).
 * ------------------------------------------------------------------------------------------ */
template <int NS>
SSG_DEVFN ssg_ext_res_t wv_extend2(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target,
                                   int w, int end_bonus, int zdrop, int h0, unsigned long long *cells)
{
	const int lane = wv_lane();
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int H[NS], E[NS], qc[NS];
	int i, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
	/* first row + query codes */
	SSG_UNROLL for (int s = 0; s < NS; ++s) {
		int j = lane + 64 * s;
		qc[s] = j < qlen ? sq_at(query, j) : 4;
		int v1 = h0 - oe_ins;                     /* eh[1].h before clamping */
		int vj = v1 - (j - 1) * e_ins, vjm1 = vj + e_ins;
		int h;
		if (j == 0) h = h0;
		else if (j == 1) h = v1 > 0 ? v1 : 0;
		else h = (j <= qlen && vjm1 > e_ins) ? vj : 0;
		H[s] = h; E[s] = 0;
	}
	{	/* band clamp */
		int mx = 0;
		for (int k = 0; k < 25; ++k) mx = mx > opt.mat[k] ? mx : opt.mat[k];
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
	for (i = 0; i < tlen; ++i) {
		int h1_init, m = 0, mj = -1, h_last;
		const int tb = sq_at(target, i);
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
		else h1_init = 0;
		if (beg >= end) { /* empty row: eh[end] = {h1_init, 0}; only the to-end bookkeeping can change */
			SSG_UNROLL for (int s = 0; s < NS; ++s) if (lane + 64 * s == end) { H[s] = h1_init; E[s] = 0; }
			if (beg == qlen) { max_ie = gscore > h1_init ? max_ie : i; gscore = gscore > h1_init ? gscore : h1_init; }
			break; /* m == 0 */
		}
		ncell += (unsigned long long)(end - beg);
		int hrow[NS], carry = SSG_NEG;
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane + 64 * s;
			const bool act = j >= beg && j < end;
			int M = H[s], e = E[s];
			M = M ? M + opt.mat[tb * 5 + qc[s]] : 0;
			int t = M - oe_ins; t = t > 0 ? t : 0;
			int g = act ? t + j * e_ins : SSG_NEG;
			int P = wv_scan_max(g);                 /* inclusive over lanes of this slot */
			P = P > carry ? P : carry;
			int Pm1 = wv_shfl(P, lane - 1);          /* P of column j-1 */
			Pm1 = lane == 0 ? carry : Pm1;
			carry = wv_shfl(P, 63);
			int f = j == beg ? 0 : Pm1 - (j - 1) * e_ins;
			int h = M > e ? M : e;
			h = h > f ? h : f;
			hrow[s] = act ? h : 0;
			if (act) {
				t = M - oe_del; t = t > 0 ? t : 0;
				e -= e_del; e = e > t ? e : t;
				E[s] = e;
			}
		}
		{	/* row maximum and the LAST column attaining it (upstream: mj = m > h ? mj : j) */
			int ml = -1, cj = -1;
			SSG_UNROLL for (int s = 0; s < NS; ++s) { const int j = lane + 64 * s; if (j >= beg && j < end) ml = ml > hrow[s] ? ml : hrow[s]; }
			m = wv_max(ml);
			SSG_UNROLL for (int s = 0; s < NS; ++s) { const int j = lane + 64 * s; if (j >= beg && j < end && hrow[s] == m) cj = j; }
			mj = wv_max(cj);
		}
		/* eh[j].h <- H(i,j-1) for j in (beg,end]; eh[beg].h <- h1_init; eh[end].e <- 0 */
		int sh[NS];
		wv_shift_cols<NS>(hrow, sh, 0);
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane + 64 * s;
			if (j == beg) H[s] = h1_init;
			else if (j > beg && j <= end) H[s] = sh[s];
			if (j == end) E[s] = 0;
		}
		{	/* h1 after the loop = H(i,end-1) */
			int v = 0;
			SSG_UNROLL for (int s = 0; s < NS; ++s) { int b = wv_shfl(hrow[s], (end - 1) & 63); if (((end - 1) >> 6) == s) v = b; }
			h_last = v;
		}
		if (end == qlen) { max_ie = gscore > h_last ? max_ie : i; gscore = gscore > h_last ? gscore : h_last; }
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			max_off = max_off > iabs(mj - i) ? max_off : iabs(mj - i);
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
		}
		/* trim the band on the freshly written eh[] */
		int nbeg = end, jlast;
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane + 64 * s;
			unsigned long long mb = wv_ballot((H[s] != 0 || E[s] != 0) && j >= beg && j < end);
			if (mb && nbeg == end) nbeg = 64 * s + __ffsll(mb) - 1;
		}
		jlast = nbeg - 1;
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane + 64 * s;
			unsigned long long me = wv_ballot((H[s] != 0 || E[s] != 0) && j >= nbeg && j <= end);
			if (me) jlast = 64 * s + 63 - __clzll(me);
		}
		beg = nbeg;
		end = jlast + 2 < qlen ? jlast + 2 : qlen;
	}
	if (cells) *cells += ncell;
	ssg_ext_res_t r; r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

SSG_DEVFN ssg_ext_res_t wv_extend2_any(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target,
                                       int w, int end_bonus, int zdrop, int h0, unsigned long long *cells)
{	/* qlen+1 columns are needed (eh[qlen]) */
	if (qlen < 64)  return wv_extend2<1>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
	if (qlen < 128) return wv_extend2<2>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
	if (qlen < 192) return wv_extend2<3>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
	return wv_extend2<4>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
}

/* ------------------------------------------------------------------------------------------
 * upstream ksw_global2.  z (backtrack bytes, tlen*n_col) may be null for score-only.
 * Returns the score (uniform).  The backtrace itself is done by ssg_global_backtrace (lane 0).
 * ------------------------------------------------------------------------------------------ */
#define SSG_MINUS_INF (-0x40000000)

template <int NS>
SSG_DEVFN int wv_global2(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target, int w, uint8_t *z, unsigned long long *cells)
{
	const int lane = wv_lane();
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	int H[NS], E[NS], qc[NS];
	SSG_UNROLL for (int s = 0; s < NS; ++s) {
		int j = lane + 64 * s;
		qc[s] = j < qlen ? sq_at(query, j) : 4;
		if (j == 0) { H[s] = 0; E[s] = SSG_MINUS_INF; }
		else if (j <= qlen && j <= w) { H[s] = -(o_ins + e_ins * j); E[s] = SSG_MINUS_INF; }
		else { H[s] = E[s] = SSG_MINUS_INF; }
	}
	unsigned long long ncell = 0;
	for (int i = 0; i < tlen; ++i) {
		const int tb = sq_at(target, i);
		const int beg = i > w ? i - w : 0;
		const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : SSG_MINUS_INF;
		int hrow[NS], carry = SSG_MINUS_INF * 2;
		if (end > beg) ncell += (unsigned long long)(end - beg);
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane + 64 * s;
			const bool act = j >= beg && j < end;
			int m = H[s] + opt.mat[tb * 5 + qc[s]], e = E[s];
			int g = act ? (m - oe_ins) + j * e_ins : SSG_MINUS_INF * 2;
			int P = wv_scan_max(g);
			P = P > carry ? P : carry;
			int Pm1 = wv_shfl(P, lane - 1);
			Pm1 = lane == 0 ? carry : Pm1;
			carry = wv_shfl(P, 63);
			/* F(i,j): decayed initial -inf, or the best opening to the left */
			int f = SSG_MINUS_INF - (j - beg) * e_ins;
			if (j > beg) { int fs = Pm1 - (j - 1) * e_ins; f = f > fs ? f : fs; }
			uint8_t d = m >= e ? 0 : 1;
			int h = m >= e ? m : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			hrow[s] = h;
			if (act) {
				int t = m - oe_del;
				e -= e_del;
				d |= e > t ? 1 << 2 : 0;
				e = e > t ? e : t;
				E[s] = e;
				t = m - oe_ins;
				int f2 = f - e_ins;
				d |= f2 > t ? 2 << 4 : 0;
				if (z) z[(long)i * n_col + (j - beg)] = d;
			}
		}
		int sh[NS];
		wv_shift_cols<NS>(hrow, sh, 0);
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane + 64 * s;
			if (end > beg) {
				if (j == beg) H[s] = h1_init;
				else if (j > beg && j <= end) H[s] = sh[s];
			} else if (j == end) H[s] = h1_init;
			if (j == end) E[s] = SSG_MINUS_INF;
		}
	}
	if (cells) *cells += ncell;
	int score = 0;
	SSG_UNROLL for (int s = 0; s < NS; ++s) { int b = wv_shfl(H[s], qlen & 63); if ((qlen >> 6) == s) score = b; }
	return score;
}

SSG_DEVFN int wv_global2_any(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target, int w, uint8_t *z, unsigned long long *cells)
{
	if (qlen < 64)  return wv_global2<1>(opt, qlen, query, tlen, target, w, z, cells);
	if (qlen < 128) return wv_global2<2>(opt, qlen, query, tlen, target, w, z, cells);
	if (qlen < 192) return wv_global2<3>(opt, qlen, query, tlen, target, w, z, cells);
	return wv_global2<4>(opt, qlen, query, tlen, target, w, z, cells);
}

/* upstream ksw_global2 backtrace; single lane.  cigar[] gets ops in forward order; returns n_cigar
 * (ops beyond `cap` are counted but not stored). */
SSG_DEVFN int ssg_global_backtrace(const uint8_t *z, int qlen, int tlen, int w, uint32_t *cigar, int cap)
{
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	int n = 0, which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
	#define SSG_PUSH(op, len) do { if (n == 0 || (op) != (int)(cigar[(n - 1 < cap ? n - 1 : cap - 1)] & 0xf) || n > cap) { if (n < cap) cigar[n] = (uint32_t)(len) << 4 | (op); ++n; } else cigar[n-1] += (uint32_t)(len) << 4; } while (0)
	while (i >= 0 && k >= 0) {
		which = z[(long)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
		if (which == 0) { SSG_PUSH(0, 1); --i; --k; }
		else if (which == 1) { SSG_PUSH(2, 1); --i; }
		else { SSG_PUSH(1, 1); --k; }
	}
	if (i >= 0) SSG_PUSH(2, i + 1);
	if (k >= 0) SSG_PUSH(1, k + 1);
	#undef SSG_PUSH
	int m = n < cap ? n : cap;
	for (i = 0; i < m >> 1; ++i) { uint32_t t = cigar[i]; cigar[i] = cigar[m-1-i]; cigar[m-1-i] = t; }
	return n;
}

/* ------------------------------------------------------------------------------------------
 * Local alignment with upstream ksw_u8/ksw_i16's observable contract (see oracle/orc_ksw.c):
 * the query is padded with zero-scoring columns to slen*p; te = first row reaching the final
 * maximum, qe = smallest column holding it; b[] = collapsed row maxima >= minsc.
 * bscratch: per-wave global scratch for b[] (>= tlen entries).
 * ------------------------------------------------------------------------------------------ */
struct ssg_sw1_t { int score, te, qe, score2, te2; };

template <int NS>
SSG_DEVFN ssg_sw1_t wv_local(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target,
                             int p, int minsc, int endsc, unsigned long long *bscratch, unsigned long long *cells)
{
	const int lane = wv_lane();
	const int e_del = opt.e_del, e_ins = opt.e_ins, oe_del = opt.o_del + e_del, oe_ins = opt.o_ins + e_ins;
	const int slen = (qlen + p - 1) / p, qp = slen * p;
	int H[NS], E[NS], HM[NS], qc[NS];
	int gmax = 0, te = -1, n_b = 0, maxsc = 0, last_sc = 0, last_row = -2;
	for (int k = 0; k < 25; ++k) maxsc = maxsc > opt.mat[k] ? maxsc : opt.mat[k];
	SSG_UNROLL for (int s = 0; s < NS; ++s) { int j = lane + 64 * s; qc[s] = j < qlen ? sq_at(query, j) : 5; H[s] = E[s] = HM[s] = 0; }
	int i;
	for (i = 0; i < tlen; ++i) {
		const int tb = sq_at(target, i);
		int diag[NS], hrow[NS], carry = SSG_NEG, imax = 0;
		wv_shift_cols<NS>(H, diag, 0);
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane + 64 * s;
			const bool act = j < qp;
			int sc = qc[s] < 5 ? opt.mat[tb * 5 + qc[s]] : 0;
			int hn = diag[s] + sc, e = E[s];
			hn = hn > e ? hn : e; hn = hn > 0 ? hn : 0;
			int g = act ? (hn - oe_ins) + j * e_ins : SSG_NEG;
			int P = wv_scan_max(g);
			P = P > carry ? P : carry;
			int Pm1 = wv_shfl(P, lane - 1);
			Pm1 = lane == 0 ? carry : Pm1;
			carry = wv_shfl(P, 63);
			int f = j == 0 ? 0 : Pm1 - (j - 1) * e_ins; f = f > 0 ? f : 0;
			int h = hn > f ? hn : f;
			hrow[s] = act ? h : 0;
			e -= e_del; { int t = h - oe_del; e = e > t ? e : t; } e = e > 0 ? e : 0;
			if (act) E[s] = e;
			int rm = wv_max(act ? h : 0);
			imax = imax > rm ? imax : rm;
		}
		SSG_UNROLL for (int s = 0; s < NS; ++s) H[s] = hrow[s];
		if (imax >= minsc) { /* b[]: collapse runs of adjacent rows, keep the entry in registers */
			if (n_b == 0 || last_row + 1 != i) { last_sc = imax; last_row = i; if (lane == 0) bscratch[n_b] = (unsigned long long)imax << 32 | (unsigned)i; ++n_b; }
			else if (last_sc < imax) { last_sc = imax; last_row = i; if (lane == 0) bscratch[n_b - 1] = (unsigned long long)imax << 32 | (unsigned)i; }
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			SSG_UNROLL for (int s = 0; s < NS; ++s) HM[s] = hrow[s];
			if (gmax >= endsc) break;
		}
	}
	if (cells) *cells += (unsigned long long)(i < tlen ? i + 1 : tlen) * qlen;
	ssg_sw1_t r; r.score = gmax; r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1;
	{	/* smallest padded column holding the row maximum of Hmax */
		int mx = -1;
		SSG_UNROLL for (int s = 0; s < NS; ++s) { int j = lane + 64 * s; int v = j < qp ? HM[s] : -1; v = wv_max(v); mx = mx > v ? mx : v; }
		int best = 1 << 30;
		SSG_UNROLL for (int s = 0; s < NS; ++s) { int j = lane + 64 * s; int c = (j < qp && HM[s] == mx) ? j : (1 << 30); c = wv_min(c); best = best < c ? best : c; }
		r.qe = best;
	}
	ssg_wave_memsync();
	if (n_b) {
		int k = (r.score + maxsc - 1) / maxsc, low = te - k, high = te + k;
		int bs = -1, bi = 1 << 30;
		for (int t = lane; t < n_b; t += 64) {
			unsigned long long v = bscratch[t]; int e = (int)(uint32_t)v, sc = (int)(v >> 32);
			if ((e < low || e > high) && sc > bs) { bs = sc; bi = t; }
		}
		int gs = wv_max(bs);
		int gi = wv_min(bs == gs ? bi : (1 << 30));
		if (gs > -1) { r.score2 = gs; r.te2 = (int)(uint32_t)bscratch[gi]; }
	}
	return r;
}

/* upstream ksw_align2 (forward pass, then the reversed pass for the start when KSW_XSTART) */
#define SSG_KSW_XBYTE  0x10000
#define SSG_KSW_XSTOP  0x20000
#define SSG_KSW_XSUBO  0x40000
#define SSG_KSW_XSTART 0x80000

template <int NS>
SSG_DEVFN ssg_kswr_t wv_align2_t(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target, int xtra,
                                 unsigned long long *bscratch, unsigned long long *cells)
{
	ssg_kswr_t r; r.tb = r.qb = -1;
	const int p = (xtra & SSG_KSW_XBYTE) ? 16 : 8;
	const int minsc = (xtra & SSG_KSW_XSUBO) ? xtra & 0xffff : 0x10000;
	const int endsc = (xtra & SSG_KSW_XSTOP) ? xtra & 0xffff : 0x10000;
	ssg_sw1_t f = wv_local<NS>(opt, qlen, query, tlen, target, p, minsc, endsc, bscratch, cells);
	r.score = f.score; r.te = f.te; r.qe = f.qe; r.score2 = f.score2; r.te2 = f.te2;
	if ((xtra & SSG_KSW_XSTART) == 0 || ((xtra & SSG_KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
	/* reverse both prefixes: walk them backwards from (qe, te) */
	ssg_seqv_t rq = { query.p + query.dir * r.qe, -query.dir }, rt = { target.p + target.dir * r.te, -target.dir };
	ssg_sw1_t rr = wv_local<NS>(opt, r.qe + 1, rq, r.te + 1, rt, p, 0x10000, r.score, bscratch, cells);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	return r;
}
SSG_DEVFN ssg_kswr_t wv_align2(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target, int xtra,
                               unsigned long long *bscratch, unsigned long long *cells)
{
	int qp = ((qlen + 7) / 8) * 8; if (xtra & SSG_KSW_XBYTE) qp = ((qlen + 15) / 16) * 16;
	if (qp <= 64)  return wv_align2_t<1>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
	if (qp <= 128) return wv_align2_t<2>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
	if (qp <= 192) return wv_align2_t<3>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
	return wv_align2_t<4>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
}
#endif
