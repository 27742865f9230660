/*
 * k_sw.h -- one-wavefront-per-job Smith-Waterman primitives for gfx950 (SURVEY.md 8a rows a7,
 * a10, a12): upstream ksw_extend2, ksw_global2 (+backtrace) and ksw_align2's contract.
 *
 * Layout (blocked): lane l of the wave owns the NS consecutive query columns j = l*NS + s
 * (NS <= 5, i.e. up to 320 columns: 2x150, 2x250, 2x300).  H(i-1,j-1)/E(i,j) of upstream's
 * eh[] array live in VGPRs for the whole job, exactly one register pair per column, so even
 * upstream's reads of stale eh[] entries beyond a shrunken band are reproduced for free.
 * A DP row is processed by all lanes at once:
 *   - the diagonal operand H(i-1,j-1) is the lane's own register (eh[j].h);
 *   - H(i,j-1) for the next row comes from the neighbouring column: the lane's own previous slot,
 *     or lane l-1's last slot via ONE wave shift (DPP wave_shr:1) per row;
 *   - the horizontal gap F(i,j) depends only on the row's M values (upstream opens E and F from M,
 *     not from H), so F is a max-plus prefix scan: F(i,j) = max_k<j (max(M_k-oe,0) - (j-1-k)*e).
 *     Each lane combines its NS columns locally, ONE 6-step DPP scan combines the lanes; for the
 *     textbook local recurrence the same holds with M replaced by max(M,E,0) because an
 *     F-opened-from-F term is dominated when o_ins > 0;
 *   - row maximum / arg-max are one more scan + v_readlane, band trimming is a ballot per slot, so
 *     beg/end/max/z-drop state stays wave-uniform (SGPRs).
 * The target base of row i+1 is fetched while row i is computed; substitution scores are computed
 * from (a, b) arithmetically (upstream bwa_fill_scmat: match a, mismatch -b, anything with N -1),
 * so the row loop touches no memory besides that one byte.
 * All arithmetic is int32 like upstream: results are bit-exact.  No MFMA: this is integer DP.
 */
#ifndef SSG_K_SW_H
#define SSG_K_SW_H
#include "ssg_dev.h"

#define SSG_NEG (-(1 << 29))

/* sequence accessor: base k of a byte-coded sequence walked with stride +1/-1 */
struct ssg_seqv_t { const uint8_t *p; int dir; };
SSG_DEVFN int sq_at(const ssg_seqv_t &s, int k) { return s.p[s.dir * k]; }

/* upstream bwa_fill_scmat as arithmetic: t = target code 0..3(4), q = query code 0..4, 5 = pad column */
SSG_DEVFN int ssg_sc(int a, int b, int t, int q) { return q > 4 ? 0 : (q > 3 || t > 3) ? -1 : (q == t ? a : -b); }

/* uniform position of the first / last set lane of a ballot */
SSG_DEVFN int ssg_first_lane(unsigned long long m) { return __ffsll(m) - 1; }
SSG_DEVFN int ssg_last_lane(unsigned long long m) { return 63 - __clzll(m); }

/* select v[idx] for a wave-uniform idx without dynamic register indexing */
template <int NS> SSG_DEVFN int ssg_pick(const int (&v)[NS], int idx)
{
	int r = v[0];
	SSG_UNROLL for (int s = 1; s < NS; ++s) r = idx == s ? v[s] : r;
	return r;
}

/* ------------------------------------------------------------------------------------------
 * upstream ksw_extend2.
 * ------------------------------------------------------------------------------------------ */
template <int NS>
SSG_DEVFN ssg_ext_res_t wv_extend2(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target,
                                   int w, int end_bonus, int zdrop, int h0, unsigned long long *cells)
{
	const int lane = wv_lane();
	const int sa = opt.a, sb = opt.b;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int H[NS], E[NS], qc[NS];
	int i, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
	SSG_UNROLL for (int s = 0; s < NS; ++s) { /* first row + query codes */
		const int j = lane * NS + s;
		qc[s] = j < qlen ? sq_at(query, j) : 4;
		const int v1 = h0 - oe_ins, vj = v1 - (j - 1) * e_ins, vjm1 = vj + e_ins;
		int h;
		if (j == 0) h = h0;
		else if (j == 1) h = v1 > 0 ? v1 : 0;
		else h = (j <= qlen && vjm1 > e_ins) ? vj : 0;
		H[s] = h; E[s] = 0;
	}
	{	/* band clamp */
		int mx = sa > 0 ? sa : 0;
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
	int tb = tlen > 0 ? sq_at(target, 0) : 0;
	for (i = 0; i < tlen; ++i) {
		int h1_init, m, mj, h_last;
		const int tb_next = i + 1 < tlen ? sq_at(target, i + 1) : 0;   /* in flight during this row */
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
		else h1_init = 0;
		if (beg >= end) { /* empty row: eh[end] = {h1_init, 0}; only the to-end bookkeeping can change */
			SSG_UNROLL for (int s = 0; s < NS; ++s) if (lane * NS + s == end) { H[s] = h1_init; E[s] = 0; }
			if (beg == qlen) { max_ie = gscore > h1_init ? max_ie : i; gscore = gscore > h1_init ? gscore : h1_init; }
			break; /* m == 0 */
		}
		ncell += (unsigned long long)(end - beg);
		int M[NS], p[NS], hrow[NS];
		SSG_UNROLL for (int s = 0; s < NS; ++s) { /* M(i,j) and the lane-local prefix of the F scan */
			const int j = lane * NS + s;
			const bool act = j >= beg && j < end;
			int mm = H[s];
			mm = mm ? mm + ssg_sc(sa, sb, tb, qc[s]) : 0;
			M[s] = mm;
			int t = mm - oe_ins; t = t > 0 ? t : 0;
			const int g = act ? t + j * e_ins : SSG_NEG;
			p[s] = s ? (p[s-1] > g ? p[s-1] : g) : g;
		}
		const int X = wv_prev(wv_scan_max(p[NS-1]), SSG_NEG);    /* best opening in the lanes to the left */
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane * NS + s;
			const bool act = j >= beg && j < end;
			const int Pm1 = s ? (X > p[s-1] ? X : p[s-1]) : X;   /* prefix up to column j-1 */
			const int f = j == beg ? 0 : Pm1 - (j - 1) * e_ins;
			int e = E[s], h = M[s] > e ? M[s] : e;
			h = h > f ? h : f;
			hrow[s] = act ? h : 0;
			if (act) {
				int t = M[s] - oe_del; t = t > 0 ? t : 0;
				e -= e_del; e = e > t ? e : t;
				E[s] = e;
			}
		}
		{	/* row maximum and the LAST column attaining it (upstream: mj = m > h ? mj : j) */
			int ml = -1, cj = -1;
			SSG_UNROLL for (int s = 0; s < NS; ++s) { const int j = lane * NS + s; if (j >= beg && j < end) ml = ml > hrow[s] ? ml : hrow[s]; }
			m = wv_max(ml);
			SSG_UNROLL for (int s = 0; s < NS; ++s) { const int j = lane * NS + s; if (j >= beg && j < end && hrow[s] == m) cj = j; }
			mj = wv_max(cj);
		}
		/* eh[j].h <- H(i,j-1) for j in (beg,end]; eh[beg].h <- h1_init; eh[end].e <- 0 */
		const int up = wv_prev(hrow[NS-1], 0);
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane * NS + s;
			const int hp = s ? hrow[s-1] : up;
			if (j == beg) H[s] = h1_init;
			else if (j > beg && j <= end) H[s] = hp;
			if (j == end) E[s] = 0;
		}
		h_last = wv_get(ssg_pick<NS>(hrow, (end - 1) % NS), (end - 1) / NS);   /* h1 after the loop = H(i,end-1) */
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
			const int j = lane * NS + s;
			const unsigned long long mb = wv_ballot((H[s] != 0 || E[s] != 0) && j >= beg && j < end);
			if (mb) { const int c = ssg_first_lane(mb) * NS + s; nbeg = nbeg < c ? nbeg : c; }
		}
		jlast = nbeg - 1;
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane * NS + s;
			const unsigned long long me = wv_ballot((H[s] != 0 || E[s] != 0) && j >= nbeg && j <= end);
			if (me) { const int c = ssg_last_lane(me) * NS + s; jlast = jlast > c ? jlast : c; }
		}
		beg = nbeg;
		end = jlast + 2 < qlen ? jlast + 2 : qlen;
		tb = tb_next;
	}
	if (cells) *cells += ncell;
	ssg_ext_res_t r; r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

/* WIDE: the kernel instance serves reads of 256..319 bases too (a fifth register column per lane).  A template parameter of every function on
 * the way down from the kernel, because its mere presence costs the narrow reads: inlined it raised the register pressure of ssg_k_chain2aln
 * (30.5 -> 38.8 ms on 2x150), out of line the call alone did (39.9 ms; measured in round 5).  The host launches the WIDE instance of a kernel
 * only for a batch with a read above 255 bases. */
template <bool WIDE>
SSG_DEVFN ssg_ext_res_t wv_extend2_any(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target,
                                       int w, int end_bonus, int zdrop, int h0, unsigned long long *cells)
{	/* qlen+1 columns are needed (eh[qlen]) */
	if (qlen < 64)  return wv_extend2<1>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
	if (qlen < 128) return wv_extend2<2>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
	if (qlen < 192) return wv_extend2<3>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
	if (!WIDE || qlen < 256) return wv_extend2<4>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
	return wv_extend2<WIDE ? 5 : 4>(opt, qlen, query, tlen, target, w, end_bonus, zdrop, h0, cells);
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
	const int sa = opt.a, sb = opt.b;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	const int NEG2 = SSG_MINUS_INF + SSG_MINUS_INF / 2;   /* below every value the recurrence can produce */
	int H[NS], E[NS], qc[NS];
	SSG_UNROLL for (int s = 0; s < NS; ++s) {
		const int j = lane * NS + s;
		qc[s] = j < qlen ? sq_at(query, j) : 4;
		if (j == 0) { H[s] = 0; E[s] = SSG_MINUS_INF; }
		else if (j <= qlen && j <= w) { H[s] = -(o_ins + e_ins * j); E[s] = SSG_MINUS_INF; }
		else { H[s] = E[s] = SSG_MINUS_INF; }
	}
	unsigned long long ncell = 0;
	int tb = tlen > 0 ? sq_at(target, 0) : 0;
	for (int i = 0; i < tlen; ++i) {
		const int tb_next = i + 1 < tlen ? sq_at(target, i + 1) : 0;
		const int beg = i > w ? i - w : 0;
		const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : SSG_MINUS_INF;
		int M[NS], p[NS], hrow[NS];
		if (end > beg) ncell += (unsigned long long)(end - beg);
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane * NS + s;
			const bool act = j >= beg && j < end;
			M[s] = H[s] + ssg_sc(sa, sb, tb, qc[s]);
			const int g = act ? (M[s] - oe_ins) + j * e_ins : NEG2;
			p[s] = s ? (p[s-1] > g ? p[s-1] : g) : g;
		}
		const int X = wv_prev(wv_scan_max(p[NS-1]), NEG2);
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane * NS + s;
			const bool act = j >= beg && j < end;
			const int Pm1 = s ? (X > p[s-1] ? X : p[s-1]) : X;
			/* F(i,j): decayed initial -inf, or the best opening to the left */
			int f = SSG_MINUS_INF - (j - beg) * e_ins;
			if (j > beg) { const int fs = Pm1 - (j - 1) * e_ins; f = f > fs ? f : fs; }
			const int m = M[s];
			int e = E[s];
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
				const int f2 = f - e_ins;
				d |= f2 > t ? 2 << 4 : 0;
				if (z) z[(long)i * n_col + (j - beg)] = d;
			}
		}
		const int up = wv_prev(hrow[NS-1], 0);
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane * NS + s;
			const int hp = s ? hrow[s-1] : up;
			if (end > beg) {
				if (j == beg) H[s] = h1_init;
				else if (j > beg && j <= end) H[s] = hp;
			} else if (j == end) H[s] = h1_init;
			if (j == end) E[s] = SSG_MINUS_INF;
		}
		tb = tb_next;
	}
	if (cells) *cells += ncell;
	return wv_get(ssg_pick<NS>(H, qlen % NS), qlen / NS);
}

template <bool WIDE>
SSG_DEVFN int wv_global2_any(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target, int w, uint8_t *z, unsigned long long *cells)
{
	if (qlen < 64)  return wv_global2<1>(opt, qlen, query, tlen, target, w, z, cells);
	if (qlen < 128) return wv_global2<2>(opt, qlen, query, tlen, target, w, z, cells);
	if (qlen < 192) return wv_global2<3>(opt, qlen, query, tlen, target, w, z, cells);
	if (!WIDE || qlen < 256) return wv_global2<4>(opt, qlen, query, tlen, target, w, z, cells);
	return wv_global2<WIDE ? 5 : 4>(opt, qlen, query, tlen, target, w, z, cells);
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
                             int p_, int minsc, int endsc, unsigned long long *bscratch, unsigned long long *cells)
{
	const int lane = wv_lane();
	const int sa = opt.a, sb = opt.b;
	const int e_del = opt.e_del, e_ins = opt.e_ins, oe_del = opt.o_del + e_del, oe_ins = opt.o_ins + e_ins;
	const int slen = (qlen + p_ - 1) / p_, qp = slen * p_;
	int H[NS], E[NS], HM[NS], qc[NS];
	int gmax = 0, te = -1, n_b = 0, last_sc = 0, last_row = -2;
	const int maxsc = sa > 0 ? sa : 0;
	/* per column, in registers: the substitution score against a target base equal to / different from / N (ssg_sc's cases), and the
	 * column's constants of the F recurrence (an inactive padded column gets g = -inf through its constant) */
	int sce[NS], scd[NS], scn[NS], cg[NS], cf[NS];
	SSG_UNROLL for (int s = 0; s < NS; ++s) {
		const int j = lane * NS + s;
		qc[s] = j < qlen ? sq_at(query, j) : 5; H[s] = E[s] = HM[s] = 0;
		sce[s] = ssg_sc(sa, sb, qc[s] < 4 ? qc[s] : 0, qc[s]); scd[s] = ssg_sc(sa, sb, qc[s] < 4 ? qc[s] ^ 1 : 0, qc[s]); scn[s] = ssg_sc(sa, sb, 4, qc[s]);
		cg[s] = j < qp ? j * e_ins - oe_ins : SSG_NEG;
		cf[s] = (j - 1) * e_ins;
	}
	/* target bases: lane l holds base 64c + l of the current chunk of 64 rows (one coalesced load per 64 rows, the next chunk in flight
	 * meanwhile); a row reads its base with v_readlane into a scalar */
	int i, tch = lane < tlen ? sq_at(target, lane) : 0, tchn = 0;
	for (i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) { if (i) tch = tchn; const int k = i + 64 + lane; tchn = k < tlen ? sq_at(target, k) : 0; }
		const int tb = wv_get(tch, i & 63);
		int hn[NS], p[NS], hrow[NS];
		const int up = wv_prev(H[NS-1], 0);          /* H(i-1, j-1) for the lane's first column */
		const bool tn = tb > 3;
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int sc = tn ? scn[s] : qc[s] == tb ? sce[s] : scd[s];
			int v = (s ? H[s-1] : up) + sc;
			const int e = E[s];
			v = v > e ? v : e; v = v > 0 ? v : 0;
			hn[s] = v;
			const int g = v + cg[s];
			p[s] = s ? (p[s-1] > g ? p[s-1] : g) : g;
		}
		const int X = wv_prev(wv_scan_max(p[NS-1]), SSG_NEG);
		int ml = 0;
		SSG_UNROLL for (int s = 0; s < NS; ++s) {
			const int j = lane * NS + s;
			const bool act = j < qp;
			const int Pm1 = s ? (X > p[s-1] ? X : p[s-1]) : X;
			/* F = max(Pm1 - (j-1) e_ins, 0) (column 0: -inf from X); hn >= 0 already, so h = max(hn, F) needs no clamp of F */
			const int f = Pm1 - cf[s];
			const int h = hn[s] > f ? hn[s] : f;
			hrow[s] = act ? h : 0;
			int e = E[s] - e_del; { const int t = hrow[s] - oe_del; e = e > t ? e : t; } e = e > 0 ? e : 0;
			E[s] = e;                                  /* a padded column stays at 0 */
			ml = ml > hrow[s] ? ml : hrow[s];
		}
		SSG_UNROLL for (int s = 0; s < NS; ++s) H[s] = hrow[s];
		/* the row maximum matters only when it reaches minsc (b[]) or exceeds the running maximum: one compare + ballot on most rows */
		const int need = minsc < gmax + 1 ? minsc : gmax + 1;
		if (wv_ballot(ml >= need) == 0) continue;
		const int imax = wv_max(ml);
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
		int mx = -1, best = 1 << 30;
		SSG_UNROLL for (int s = 0; s < NS; ++s) { const int j = lane * NS + s; const int v = j < qp ? HM[s] : -1; mx = mx > v ? mx : v; }
		mx = wv_max(mx);
		SSG_UNROLL for (int s = 0; s < NS; ++s) { const int j = lane * NS + s; const int c = (j < qp && HM[s] == mx) ? j : (1 << 30); best = best < c ? best : c; }
		r.qe = wv_min(best);
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

/* second half of upstream ksw_align2 (KSW_XSTART): the start of the alignment from the reversed prefixes, walked backwards from (qe, te).
 * r holds the forward pass (score, te, qe); only target rows 0..te are read. */
template <int NS>
SSG_DEVFN void wv_align2_rev_t(const ssg_mem_opt_t &opt, ssg_seqv_t query, ssg_seqv_t target, int xtra, ssg_kswr_t &r,
                               unsigned long long *bscratch, unsigned long long *cells)
{
	const int p = (xtra & SSG_KSW_XBYTE) ? 16 : 8;
	r.tb = r.qb = -1;
	if ((xtra & SSG_KSW_XSTART) == 0 || ((xtra & SSG_KSW_XSUBO) && r.score < (xtra & 0xffff))) return;
	ssg_seqv_t rq = { query.p + query.dir * r.qe, -query.dir }, rt = { target.p + target.dir * r.te, -target.dir };
	ssg_sw1_t rr = wv_local<NS>(opt, r.qe + 1, rq, r.te + 1, rt, p, 0x10000, r.score, bscratch, cells);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
}
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
	wv_align2_rev_t<NS>(opt, query, target, xtra, r, bscratch, cells);
	return r;
}
/* does ksw_align2 run its reverse pass after a forward pass that ended with this score? */
SSG_DEVFN bool ssg_align2_has_rev(int xtra, int score) { return (xtra & SSG_KSW_XSTART) != 0 && !((xtra & SSG_KSW_XSUBO) && score < (xtra & 0xffff)); }
SSG_DEVFN int ssg_align2_qp(int qlen, int xtra) { return (xtra & SSG_KSW_XBYTE) ? ((qlen + 15) / 16) * 16 : ((qlen + 7) / 8) * 8; }
template <bool WIDE>
SSG_DEVFN ssg_kswr_t wv_align2(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, int tlen, ssg_seqv_t target, int xtra,
                               unsigned long long *bscratch, unsigned long long *cells)
{
	const int qp = ssg_align2_qp(qlen, xtra);
	if (qp <= 64)  return wv_align2_t<1>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
	if (qp <= 128) return wv_align2_t<2>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
	if (qp <= 192) return wv_align2_t<3>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
	if (!WIDE || qp <= 256) return wv_align2_t<4>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
	return wv_align2_t<WIDE ? 5 : 4>(opt, qlen, query, tlen, target, xtra, bscratch, cells);
}
/* ksw_align2 whose forward pass (r.score, te, qe, score2, te2) was computed elsewhere (k_mswlane.h): the reverse pass in the column layout
 * the whole call would have used */
template <bool WIDE>
SSG_DEVFN void wv_align2_rev(const ssg_mem_opt_t &opt, int qlen, ssg_seqv_t query, ssg_seqv_t target, int xtra, ssg_kswr_t &r,
                             unsigned long long *bscratch, unsigned long long *cells)
{
	const int qp = ssg_align2_qp(qlen, xtra);
	if (qp <= 64)  return wv_align2_rev_t<1>(opt, query, target, xtra, r, bscratch, cells);
	if (qp <= 128) return wv_align2_rev_t<2>(opt, query, target, xtra, r, bscratch, cells);
	if (qp <= 192) return wv_align2_rev_t<3>(opt, query, target, xtra, r, bscratch, cells);
	if (!WIDE || qp <= 256) return wv_align2_rev_t<4>(opt, query, target, xtra, r, bscratch, cells);
	return wv_align2_rev_t<WIDE ? 5 : 4>(opt, query, target, xtra, r, bscratch, cells);
}
#endif
