/*
 * oracle/orc_ksw.c -- CPU ORACLE (test infrastructure): Smith-Waterman kernels, scalar int32.
 *
 * Restates upstream lh3/bwa ksw.c: ksw_extend2 (banded extension from a fixed origin),
 * ksw_global2 (banded global alignment with backtrace) and ksw_align2 (local alignment used by
 * mate rescue; upstream is SSE2 striped -- here its observable contract is reproduced with a
 * scalar DP, see orc_ksw_align2).  Upstream sources are absent from /root/reference; the
 * recurrences follow SURVEY.md Appendix B.  PARITY UNPINNED (see orc.h).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "orc.h"

__thread uint64_t orc_cnt_cells;

typedef struct { int32_t h, e; } eh_t;

int orc_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                    int *_qle, int *_tle, int *_gtle, int *_gscore, int *_max_off)
{
	eh_t *eh; int8_t *qp;
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
	assert(h0 > 0);
	qp = malloc((size_t)qlen * m + 1);
	eh = calloc(qlen + 1, 8);
	for (k = i = 0; k < m; ++k) { /* query profile */
		const int8_t *p = &mat[k * m];
		for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
	}
	/* first row */
	eh[0].h = h0; eh[1].h = h0 > oe_ins ? h0 - oe_ins : 0;
	for (j = 2; j <= qlen && eh[j-1].h > e_ins; ++j) eh[j].h = eh[j-1].h - e_ins;
	/* clamp the band */
	k = m * m;
	for (i = 0, max = 0; i < k; ++i) max = max > mat[i] ? max : mat[i];
	max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
	max_ins = max_ins > 1 ? max_ins : 1;
	w = w < max_ins ? w : max_ins;
	max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
	max_del = max_del > 1 ? max_del : 1;
	w = w < max_del ? w : max_del;
	/* DP */
	max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
	beg = 0; end = qlen;
	for (i = 0; i < tlen; ++i) {
		int t, f = 0, h1, mm = 0, mj = -1;
		int8_t *q = &qp[target[i] * qlen];
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
		else h1 = 0;
		for (j = beg; j < end; ++j) {
			/* eh[j] = {H(i-1,j-1), E(i,j)}, f = F(i,j), h1 = H(i,j-1) */
			eh_t *p = &eh[j];
			int h, M = p->h, e = p->e;
			p->h = h1;
			M = M ? M + q[j] : 0; /* a zero H cannot restart through the diagonal */
			h = M > e ? M : e;
			h = h > f ? h : f;
			h1 = h;
			mj = mm > h ? mj : j;
			mm = mm > h ? mm : h;
			t = M - oe_del; t = t > 0 ? t : 0;
			e -= e_del; e = e > t ? e : t;
			p->e = e;
			t = M - oe_ins; t = t > 0 ? t : 0;
			f -= e_ins; f = f > t ? f : t;
		}
		if (end > beg) orc_cnt_cells += end - beg;
		eh[end].h = h1; eh[end].e = 0;
		if (j == qlen) {
			max_ie = gscore > h1 ? max_ie : i;
			gscore = gscore > h1 ? gscore : h1;
		}
		if (mm == 0) break;
		if (mm > max) {
			max = mm; max_i = i; max_j = mj;
			max_off = max_off > abs(mj - i) ? max_off : abs(mj - i);
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) {
				if (max - mm - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
			} else {
				if (max - mm - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
			}
		}
		for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j);
		beg = j;
		for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j);
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	free(eh); free(qp);
	if (_qle) *_qle = max_j + 1;
	if (_tle) *_tle = max_i + 1;
	if (_gtle) *_gtle = max_ie + 1;
	if (_gscore) *_gscore = gscore;
	if (_max_off) *_max_off = max_off;
	return max;
}

#define MINUS_INF -0x40000000

static inline uint32_t *push_cigar(int *n_cigar, int *m_cigar, uint32_t *cigar, int op, int len)
{
	if (*n_cigar == 0 || op != (int)(cigar[(*n_cigar) - 1] & 0xf)) {
		if (*n_cigar == *m_cigar) { *m_cigar = *m_cigar ? (*m_cigar) << 1 : 4; cigar = realloc(cigar, (*m_cigar) << 2); }
		cigar[(*n_cigar)++] = len << 4 | op;
	} else cigar[(*n_cigar) - 1] += len << 4;
	return cigar;
}

int orc_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t **cigar_)
{
	eh_t *eh; int8_t *qp;
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, score, n_col;
	uint8_t *z; /* per cell: f<<4 | e<<2 | h */
	if (n_cigar_) *n_cigar_ = 0;
	n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	z = n_cigar_ && cigar_ ? malloc((size_t)n_col * tlen + 1) : 0;
	qp = malloc((size_t)qlen * m + 1);
	eh = calloc(qlen + 1, 8);
	for (k = i = 0; k < m; ++k) {
		const int8_t *p = &mat[k * m];
		for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
	}
	eh[0].h = 0; eh[0].e = MINUS_INF;
	for (j = 1; j <= qlen && j <= w; ++j) eh[j].h = -(o_ins + e_ins * j), eh[j].e = MINUS_INF;
	for (; j <= qlen; ++j) eh[j].h = eh[j].e = MINUS_INF;
	for (i = 0; i < tlen; ++i) {
		int32_t f = MINUS_INF, h1, beg, end, t;
		int8_t *q = &qp[target[i] * qlen];
		uint8_t *zi = z ? &z[(size_t)i * n_col] : 0;
		beg = i > w ? i - w : 0;
		end = i + w + 1 < qlen ? i + w + 1 : qlen;
		h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
		for (j = beg; j < end; ++j) {
			eh_t *p = &eh[j];
			int32_t h, mm = p->h, e = p->e;
			uint8_t d;
			p->h = h1;
			mm += q[j];
			d = mm >= e ? 0 : 1;
			h = mm >= e ? mm : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			h1 = h;
			t = mm - oe_del;
			e -= e_del;
			d |= e > t ? 1 << 2 : 0;
			e = e > t ? e : t;
			p->e = e;
			t = mm - oe_ins;
			f -= e_ins;
			d |= f > t ? 2 << 4 : 0;
			f = f > t ? f : t;
			if (zi) zi[j - beg] = d;
		}
		if (end > beg) orc_cnt_cells += end - beg;
		eh[end].h = h1; eh[end].e = MINUS_INF;
	}
	score = eh[qlen].h;
	if (z) { /* backtrack */
		int n_cigar = 0, m_cigar = 0, which = 0;
		uint32_t *cigar = 0, tmp;
		i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0)      cigar = push_cigar(&n_cigar, &m_cigar, cigar, 0, 1), --i, --k;
			else if (which == 1) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, 1), --i;
			else                 cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, 1), --k;
		}
		if (i >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, i + 1);
		if (k >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, k + 1);
		for (i = 0; i < n_cigar >> 1; ++i) tmp = cigar[i], cigar[i] = cigar[n_cigar-1-i], cigar[n_cigar-1-i] = tmp;
		*n_cigar_ = n_cigar; *cigar_ = cigar;
	}
	free(eh); free(qp); free(z);
	return score;
}

/*
 * Local alignment with upstream ksw_u8/ksw_i16's observable contract, scalar:
 *  - textbook affine local SW (H>=0), E/F opened from H;
 *  - the query is padded with zero-scoring columns to slen*p (p = 16 lanes for the byte kernel,
 *    8 for the word kernel) exactly as the striped kernels do, so row maxima see pad cells;
 *  - te = first target row where the running maximum strictly increases to its final value,
 *    qe = smallest query index holding that maximum in row te;
 *  - row maxima >= minsc are collapsed over adjacent rows into the b[] list from which
 *    score2/te2 (best hit ending outside te +- ceil(score/max_match)) are taken;
 *  - the scan stops at the first row whose maximum reaches endsc.
 * Divergence from upstream (documented, unreachable with default scoring): upstream's lazy-F
 * pass does not re-open E from an F-raised H.
 */
typedef struct { int score, te, qe, score2, te2; } sw1_t;

/* no_e_from_f: E is never opened from an H that F alone raised -- MORE than upstream leaves out (its lazy-F pass skips the re-opening only for an F that crosses a
 * stripe boundary), so a call whose result is the same with and without this switch is a call on which the textbook recurrence and upstream's agree (orc_ksw_align2
 * counts the calls that differ when ORC_LAZYF_COUNT is set: bench.py's parity object, tests/test_kernels_emu.py) */
static sw1_t local_sw(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                      int o_del, int e_del, int o_ins, int e_ins, int p, int minsc, int endsc, int no_e_from_f)
{
	sw1_t r = { 0, -1, -1, -1, -1 };
	int slen = (qlen + p - 1) / p, qp = slen * p, i, j, gmax = 0, te = -1, n_b = 0, m_b = 0, maxsc = 0;
	int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int32_t *H = calloc(qp + 1, 4), *E = calloc(qp + 1, 4), *Hmax = calloc(qp + 1, 4);
	uint64_t *b = 0;
	for (i = 0; i < m * m; ++i) maxsc = maxsc > mat[i] ? maxsc : mat[i];
	for (i = 0; i < tlen; ++i) {
		int32_t diag = 0, f = 0, imax = 0;
		const int8_t *row = &mat[target[i] * m];
		for (j = 0; j < qp; ++j) {
			int32_t s = j < qlen ? row[query[j]] : 0;
			int32_t h = diag + s, e = E[j];
			diag = H[j];
			h = h > e ? h : e;
			const int32_t h_before_f = h;
			h = h > f ? h : f;
			h = h > 0 ? h : 0;
			H[j] = h;
			imax = imax > h ? imax : h;
			e -= e_del; { int32_t t = (no_e_from_f ? (h_before_f > 0 ? h_before_f : 0) : h) - oe_del; e = e > t ? e : t; } E[j] = e > 0 ? e : 0;
			f -= e_ins; { int32_t t = h - oe_ins; f = f > t ? f : t; } f = f > 0 ? f : 0;
		}
		orc_cnt_cells += qlen;
		if (imax >= minsc) {
			if (n_b == 0 || (int32_t)b[n_b-1] + 1 != i) {
				if (n_b == m_b) { m_b = m_b ? m_b << 1 : 8; b = realloc(b, 8 * m_b); }
				b[n_b++] = (uint64_t)imax << 32 | i;
			} else if ((int)(b[n_b-1] >> 32) < imax) b[n_b-1] = (uint64_t)imax << 32 | i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax, H, qp * 4);
			if (gmax >= endsc) break;
		}
	}
	r.score = gmax; r.te = te;
	{
		int max = -1, low, high;
		for (j = 0; j < qp; ++j) if (Hmax[j] > max) max = Hmax[j], r.qe = j; /* smallest index with the max */
		if (b) {
			i = (r.score + maxsc - 1) / maxsc;
			low = te - i; high = te + i;
			for (i = 0; i < n_b; ++i) {
				int e = (int32_t)b[i];
				if ((e < low || e > high) && (int)(b[i] >> 32) > r.score2) r.score2 = b[i] >> 32, r.te2 = e;
			}
		}
	}
	free(H); free(E); free(Hmax); free(b);
	return r;
}

static void revseq(int l, uint8_t *s) { for (int i = 0; i < l >> 1; ++i) { uint8_t t = s[i]; s[i] = s[l-1-i]; s[l-1-i] = t; } }

int orc_lazyf_count = -1;   /* -1: ask the environment (ORC_LAZYF_COUNT) at the first call */
uint64_t orc_lazyf_calls = 0, orc_lazyf_differ = 0;   /* ORC_LAZYF_COUNT: ksw_align2 calls looked at / whose result changes when E is never opened from an F-raised H */
static orc_kswr_t align2_mode(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, int no_e_from_f);
orc_kswr_t orc_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                          int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
	if (orc_lazyf_count < 0) orc_lazyf_count = getenv("ORC_LAZYF_COUNT") != 0;
	const int count = orc_lazyf_count;
	const orc_kswr_t r = align2_mode(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, xtra, 0);
	if (count) {
		const orc_kswr_t s = align2_mode(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, xtra, 1);
		__atomic_fetch_add(&orc_lazyf_calls, 1, __ATOMIC_RELAXED);
		if (memcmp(&r, &s, sizeof(r)) != 0) __atomic_fetch_add(&orc_lazyf_differ, 1, __ATOMIC_RELAXED);
	}
	return r;
}
static orc_kswr_t align2_mode(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                          int o_del, int e_del, int o_ins, int e_ins, int xtra, int no_e_from_f)
{	/* upstream ksw_align2 */
	orc_kswr_t r = { 0, -1, -1, -1, -1, -1, -1 };
	int p = (xtra & ORC_KSW_XBYTE) ? 16 : 8;
	int minsc = (xtra & ORC_KSW_XSUBO) ? xtra & 0xffff : 0x10000;
	int endsc = (xtra & ORC_KSW_XSTOP) ? xtra & 0xffff : 0x10000;
	sw1_t f = local_sw(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, p, minsc, endsc, no_e_from_f), rr;
	r.score = f.score; r.te = f.te; r.qe = f.qe; r.score2 = f.score2; r.te2 = f.te2;
	if ((xtra & ORC_KSW_XSTART) == 0 || ((xtra & ORC_KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
	revseq(r.qe + 1, query); revseq(r.te + 1, target);
	rr = local_sw(r.qe + 1, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, p, 0x10000, r.score, no_e_from_f);
	revseq(r.qe + 1, query); revseq(r.te + 1, target);
	if (r.score == rr.score) r.tb = r.te - rr.te, r.qb = r.qe - rr.qe;
	return r;
}
