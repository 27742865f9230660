/*
 * k_mswlane.h -- mate rescue's local Smith-Waterman ahead of the decision (upstream mem_matesw -> ksw_align2).
 *
 * mem_matesw walks a pair's anchors in order and a rescued hit changes which later windows are tried, so the DECISION to align a window is
 * sequential per pair side.  The RESULT of a window's alignment depends only on (anchor, orientation, mate): it can be computed ahead of the
 * decision, exactly as k_extlane.h does for the seed extensions.  So:
 *   ssg_k_msw_count   one lane per listed pair: anchors per side (upstream's b[i]: hits within pen_unpaired of the best) -> 4 slots per anchor
 *   ssg_k_msw_emit    one lane per anchor: the skip test of mem_matesw on the hit lists AS THEY ARE BEFORE ANY RESCUE, the window of every
 *                     orientation that passes -> a job record in the anchor's slot, a sort key (padded query length, window length)
 *   ssg_k_msw_lane    the forward pass of ksw_align2 (oracle/orc_ksw.c local_sw: score, te, qe, b[] -> score2 / te2) for every job,
 *                     L LANES PER JOB (64 / L jobs per wavefront), the DP in strips of R = 8 target rows:
 *                       - a lane owns C = qp / L consecutive query columns; per column ONE LDS word holds the strip boundary
 *                         (H of the row above the strip : 13 | E of the strip's first row : 13 | score-table offset of the query base : 5);
 *                         word (column * 64 + lane): bank = lane, conflict free.  One ds_read + one ds_write per EIGHT cells;
 *                       - the 8 rows of a strip run down a column in registers: H of the column to the left (the diagonal operand), F and the
 *                         row maximum per row, E handed down; 11 VALU per cell (bfe score, add, max3; shift-or + max for the row maximum and
 *                         its smallest column; 3 + 3 for E and F) against ~34 lane-operations per cell of the row-parallel wave form
 *                         (k_sw.h wv_local: 150 columns on 192 lane slots, two 6-step DPP scans per row);
 *                       - lanes of a job are one strip apart (lane c works on strip s - c): the right edge of a strip (H, F, row maxima,
 *                         the diagonal) moves to the next lane by ONE wave_shr:1 DPP move per value and strip;
 *                       - target rows come straight from the 2-bit pac (3 byte loads per strip), no window copy in HBM.
 *                     Results go to the job's slot; b[] lists live in a small per-workgroup scratch (rare writes).
 *   ssg_k_matesw      (k_pair.h) replays upstream's decisions per pair and takes a window's forward pass from its slot when there is one
 *                     (rb and length checked), runs only the reverse pass (KSW_XSTART) for the windows that reached min_seed_len; a window
 *                     without a slot -- a skip decision that changed because of an earlier rescue, an over-long window, scores beyond the packed cells -- goes through the wave code as before.  Results are the same
 *                     either way; only the time differs.
 * All arithmetic is int32 like upstream; DP values stay below 8191 (13-bit fields; checked on the host: qlen * a <= 8190).
 */
#ifndef SSG_K_MSWLANE_H
#define SSG_K_MSWLANE_H
#include "k_sw.h"

SSG_DEVFN int ssg_infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{	/* upstream mem_infer_dir */
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

#define SSG_ML_R 8            /* target rows per strip */
#define SSG_ML_TMAX 8192      /* longest window given to the lane kernel */

/* slot = side base + 4 * anchor + orientation.  p = 16 / 8: the padding unit of the query (KSW_XBYTE or not); xstart: KSW_XSTART */
struct ssg_msjob_t { int64_t rb, qoff; int32_t tlen, qlen, qp, minsc, is_rev, p, xstart, _pad; };                 /* 48 bytes */
struct ssg_msres_t { int64_t rb; int32_t tlen, state, score, te, qe, score2, te2, tb, qb, _pad; };   /* 48 bytes; state: 0 = not computed, 1 = forward pass, 2 = forward and reverse pass (tb, qb) */

/* the lane kernel trusts no job record: a key that names no slot, or a record with impossible lengths, is skipped (the wave code does that
 * window then); -DSSG_ML_CHECK counts them in ssg_dbg_cyc[24 + code] */
#ifdef SSG_ML_CHECK
#define SSG_ML_OK(code, cond) ((cond) ? true : (atomicAdd(&ssg_dbg_cyc[24 + (code)], 1ull), false))
#else
#define SSG_ML_OK(code, cond) (cond)
#endif
#define SSG_ML_QS(w) ((w) & 31u)
#define SSG_ML_E(w)  ((int)(((w) >> 5) & 0x1fffu))
#define SSG_ML_H(w)  ((int)((w) >> 18))
#ifdef SSG_EMU
SSG_DEVFN int ssg_sbfe5(uint32_t t, unsigned off) { return (int)(t << (27 - off)) >> 27; }
#else
SSG_DEVFN int ssg_sbfe5(uint32_t t, unsigned off) { return __builtin_amdgcn_sbfe((int)t, off, 5u); }
#endif
SSG_DEVFN int ssg_max3(int a, int b, int c) { a = a > b ? a : b; return a > c ? a : c; }

/* may this scoring / read go through the 13-bit cells and the 5-bit score table? */
SSG_DEVFN bool ssg_ml_fits(const ssg_mem_opt_t &opt, int qlen)
{
	return opt.a >= 1 && opt.a <= 15 && opt.b >= 0 && opt.b <= 16 && qlen >= 1 && qlen <= 320 && qlen * opt.a <= 8190;
}

/* target bases of 8 consecutive rows: doubled coordinates p .. p + 7 of the 2-bit pac, 4 bits per row (row 0 lowest).  Rows outside the
 * reference read as anything (the caller masks them); every byte index stays inside the pac array. */
SSG_DEVFN uint32_t ssg_ml_rows8_up(const ssg_index_view_t &ix, int64_t p)
{
	const int64_t l2 = ix.l_pac << 1, nb = (ix.l_pac >> 2) + 1;
	const bool fw = p < ix.l_pac;
	const int64_t lo = fw ? p : l2 - 8 - p;           /* lowest forward-strand position of the 8 rows */
	int64_t b = lo >> 2;
	b = b > nb - 3 ? nb - 3 : b; b = b < 0 ? 0 : b;
	const uint32_t v = (uint32_t)ix.pac[b] << 16 | (uint32_t)ix.pac[b + 1 < nb ? b + 1 : b] << 8 | (uint32_t)ix.pac[b + 2 < nb ? b + 2 : b];
	const int o0 = (int)((fw ? p : l2 - 1 - p) - (b << 2));   /* offset of row 0's base from base 4b; row r: o0 + r (forward) / o0 - r */
	uint32_t out = 0;
	SSG_UNROLL for (int r = 0; r < 8; ++r) {
		const int o = fw ? o0 + r : o0 - r;
		const uint32_t base = (o >= 0 && o < 12) ? (v >> (22 - 2 * o)) & 3u : 0u;
		out |= (fw ? base : 3u - base) << (4 * r);
	}
	return out;
}

/* REV: rows walk the doubled coordinate downwards from p (the reverse pass: p, p - 1, ...): the ascending fetch of p - 7 .. p, nibbles reversed */
template <bool REV> SSG_DEVFN uint32_t ssg_ml_rows8(const ssg_index_view_t &ix, int64_t p)
{
	if (!REV) return ssg_ml_rows8_up(ix, p);
	int64_t p0 = p - 7;                                /* rows r > p - p0 lie before the strand's first base: never real rows of a window */
	if (p0 < ix.l_pac && p >= ix.l_pac) p0 = ix.l_pac;
	if (p0 < 0) p0 = 0;
	uint32_t x = ssg_ml_rows8_up(ix, p0) << (4 * (7 - (int)(p - p0)));
	x = (x >> 16) | (x << 16); x = ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu);   /* bytes reversed */
	return ((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu);
}

/*
 * Forward pass of ksw_align2 for the jobs sorted[0 .. n_jobs), L lanes per job.  One wavefront per workgroup; workgroups take chunks of
 * 64 / L jobs from `queue`.  Dynamic LDS: (ccap + 2) * 64 words, ccap >= (largest qp) / L.  bglb: gridDim.x * (64 / L) * bcap entries; bcap >= (longest window) / 2 + 1 holds every b[] list
 * (upstream appends at most every other row).
 * A chunk's jobs must share qp (the sort key's leading field); a job whose qp differs from the chunk's first is left (state 0).
 */
template <int L, bool REV>
__global__ void __launch_bounds__(64) ssg_k_msw_lane(ssg_index_view_t ix, ssg_mem_opt_t opt, long n_jobs, const uint64_t *sorted, const ssg_msjob_t *jobs,
                               const uint8_t *seq, ssg_msres_t *res, unsigned long long *bglb, unsigned int *queue, int ccap, int bcap, unsigned long long *cells, long n_slots, long seq_bytes)
{
#ifdef SSG_EMU
	uint32_t *ml_lds = (uint32_t*)emu::dyn_lds;
#else
	extern __shared__ uint32_t ml_lds[];
#endif
	constexpr int J = 64 / L, R = SSG_ML_R;
	const int lane = wv_lane(), c = lane % L, jslot = lane / L;
	uint32_t *Lc = ml_lds + lane;
	unsigned long long *bl = bglb + ((long)blockIdx.x * J + jslot) * bcap;
	const int e_del = opt.e_del, e_ins = opt.e_ins, oe_del = opt.o_del + opt.e_del, oe_ins = opt.o_ins + opt.e_ins;
	/* score table of a target base t < 4: 5-bit fields, field q = score of query code q (0..3 bases, 4 = N, 5 = pad column) */
	uint32_t t_mis = (uint32_t)(-1 & 31) << 20;
	SSG_UNROLL for (int q = 0; q < 4; ++q) t_mis |= (uint32_t)(-opt.b & 31) << (5 * q);
	const uint32_t t_x = (uint32_t)((-opt.b ^ opt.a) & 31);
	const int maxsc = opt.a > 0 ? opt.a : 1;
	unsigned long long ncell = 0;
	for (;;) {
		const long chunk = wv_queue_pop(queue);
		if (chunk * J >= n_jobs) break;
		const long t = chunk * J + jslot;
		bool pending = t < n_jobs;
		const long slot = pending ? (long)(uint32_t)sorted[t] : 0;
		ssg_msjob_t jb; jb.rb = 0; jb.qoff = 0; jb.tlen = 0; jb.qlen = 0; jb.qp = 0; jb.minsc = 0x10000; jb.is_rev = 0; jb.p = 16; jb.xstart = 0; jb._pad = 0;
		if (pending && !SSG_ML_OK(0, slot < n_slots)) pending = false;
		if (pending) jb = jobs[slot];
		if (pending && !SSG_ML_OK(1, jb.qlen >= 1 && jb.qlen <= 320 && jb.tlen >= 1 && jb.tlen <= SSG_ML_TMAX && jb.qoff >= 0 && jb.qoff + jb.qlen <= seq_bytes && jb.rb >= 0 && jb.rb + jb.tlen <= (ix.l_pac << 1))) pending = false;
		int f_te = 0, f_qe = 0, endsc = 0x10000, q_full = jb.qlen;   /* REV: the forward pass's end (the reverse pass starts there) and its score (where it stops) */
		if (REV && pending) {	/* upstream ksw_align2: query[0..qe] and target[0..te] reversed, no b[], stop at the forward score */
			const ssg_msres_t fw = res[slot];
			if (!SSG_ML_OK(3, fw.state == 1 && fw.te >= 0 && fw.te < jb.tlen && fw.qe >= 0 && fw.qe < jb.qlen && (jb.p == 8 || jb.p == 16))) pending = false;
			else { f_te = fw.te; f_qe = fw.qe; endsc = fw.score; jb.tlen = fw.te + 1; jb.qlen = fw.qe + 1; jb.qp = (jb.qlen + jb.p - 1) / jb.p * jb.p; jb.minsc = 0x10000; }
		}
		for (;;) {	/* one pass per padded query length in the chunk (the jobs are sorted by it: one pass, two where lengths meet) */
		const unsigned long long pm = wv_ballot(pending);
		if (pm == 0) break;
		const int qp0 = wv_get(jb.qp, __ffsll(pm) - 1);
		const bool have = pending && jb.qp == qp0;
		pending = pending && !have;
		const int C = qp0 / L;
		if (C < 2 || C > ccap || C * L != qp0 || (C & 1)) continue;   /* wave-uniform; cannot happen with the host's launch parameters */
		const int tlen = have ? jb.tlen : 0;
		const int nstrip = (wv_max(tlen) + R - 1) / R;
		for (int cc = 0; cc < C + 2; ++cc) {       /* the lane's columns: query code -> table offset, H = E = 0 */
			const int j = c * C + cc;
			int code = 5;
			if (have && cc < C && j < jb.qlen) {
				const int k = REV ? f_qe - j : j;      /* column j of the reverse pass is column qe - j of the forward one */
				if (jb.is_rev) { const int b0 = seq[jb.qoff + q_full - 1 - k]; code = b0 < 4 ? 3 - b0 : 4; }
				else code = seq[jb.qoff + k];
			}
			Lc[cc * 64] = (uint32_t)(code * 5);
		}
		int hA[R], hB[R], f[R], rm[R], dg = 0;
		SSG_UNROLL for (int r = 0; r < R; ++r) { hA[r] = 0; hB[r] = 0; f[r] = 0; rm[r] = 0; }
		int gmax = 0, te = -1, qe = 0, n_b = 0, last_sc = 0, last_row = -2, done = 0;
		for (int s = 0; s < nstrip + L - 1; ++s) {
			const int ks = s - c, i0 = ks * R;
			const bool act = ks >= 0 && i0 < tlen;
			if (REV && wv_ballot(have && c == L - 1 && !done && i0 < tlen) == 0) break;   /* every job of the wave has reached its score (or its last row) */
			/* the left edge of this lane's strip: the right edge of the same strip in the lane to the left (a step ago); column -1 for the job's first lane */
			if (L > 1) {
				SSG_UNROLL for (int r = 0; r < R; ++r) { hA[r] = wv_prev(hA[r], 0); f[r] = wv_prev(f[r], 0); rm[r] = wv_prev(rm[r], 0); }
				dg = wv_prev(dg, 0);
			}
			if (L == 1 || c == 0) { SSG_UNROLL for (int r = 0; r < R; ++r) { hA[r] = 0; f[r] = 0; rm[r] = 0; } dg = 0; }
			if (wv_ballot(act) == 0) continue;
			if (act) {
				const uint32_t rows = REV ? ssg_ml_rows8<true>(ix, jb.rb + f_te - i0) : ssg_ml_rows8<false>(ix, jb.rb + i0);
				uint32_t T[R];
				SSG_UNROLL for (int r = 0; r < R; ++r) { const uint32_t tb = rows >> (4 * r) & 3u; T[r] = t_mis ^ (t_x << (5 * tb)); }
				int tag = 511 - c * C;   /* 511 - column: the larger key of two equal scores is the smaller column */
				uint32_t w0 = Lc[0], w1 = Lc[64];
				for (int cc = 0; cc < C; cc += 2) {
					const uint32_t n0 = Lc[(cc + 2) * 64], n1 = Lc[(cc + 3) * 64];   /* next pair in flight (two spare columns behind the last) */
					{	/* column cc: left neighbour in hA, result in hB */
						const unsigned qs = SSG_ML_QS(w0);
						int e = SSG_ML_E(w0), d = dg, h = 0;
						dg = SSG_ML_H(w0);
						SSG_UNROLL for (int r = 0; r < R; ++r) {
							const int m = d + ssg_sbfe5(T[r], qs);
							d = hA[r];
							h = ssg_max3(m, e, f[r]);
							hB[r] = h;
							{ const int k = h << 9 | tag; rm[r] = rm[r] > k ? rm[r] : k; }
							e = ssg_max3(e - e_del, h - oe_del, 0);
							f[r] = ssg_max3(f[r] - e_ins, h - oe_ins, 0);
						}
						Lc[cc * 64] = (uint32_t)h << 18 | (uint32_t)e << 5 | qs;
					}
					--tag;
					{	/* column cc + 1: left neighbour in hB, result in hA */
						const unsigned qs = SSG_ML_QS(w1);
						int e = SSG_ML_E(w1), d = dg, h = 0;
						dg = SSG_ML_H(w1);
						SSG_UNROLL for (int r = 0; r < R; ++r) {
							const int m = d + ssg_sbfe5(T[r], qs);
							d = hB[r];
							h = ssg_max3(m, e, f[r]);
							hA[r] = h;
							{ const int k = h << 9 | tag; rm[r] = rm[r] > k ? rm[r] : k; }
							e = ssg_max3(e - e_del, h - oe_del, 0);
							f[r] = ssg_max3(f[r] - e_ins, h - oe_ins, 0);
						}
						Lc[(cc + 1) * 64] = (uint32_t)h << 18 | (uint32_t)e << 5 | qs;
					}
					--tag;
					w0 = n0; w1 = n1;
				}
				if (c == L - 1) {	/* the rows of this strip are complete: upstream's per-row bookkeeping, in row order */
					SSG_UNROLL for (int r = 0; r < R; ++r) {
						const int i = i0 + r, imax = rm[r] >> 9;
						if (i < tlen && !done) {
							if (!REV && imax >= jb.minsc) {
								const unsigned long long pk = (unsigned long long)imax << 32 | (unsigned)i;
								if (n_b == 0 || last_row + 1 != i) { last_sc = imax; last_row = i; if (n_b < bcap && SSG_ML_OK(2, ((long)blockIdx.x * J + jslot) * bcap + n_b < (long)gridDim.x * J * bcap)) bl[n_b] = pk; ++n_b; }
								else if (last_sc < imax) { last_sc = imax; last_row = i; if (n_b <= bcap) bl[n_b - 1] = pk; }
							}
							if (imax > gmax) { gmax = imax; te = i; qe = 511 - (rm[r] & 511); if (REV && gmax >= endsc) done = 1; }
						}
					}
				}
			}
		}
		if (REV) {
			if (have && c == L - 1) {	/* upstream: the start is known when the reverse pass reaches the forward score */
				ssg_msres_t o = res[slot];
				o.state = 2; o.tb = o.qb = -1;
				if (gmax == endsc) { o.tb = f_te - te; o.qb = f_qe - qe; }
				res[slot] = o;
			}
		} else
		if (have && c == L - 1) {
			ssg_msres_t o; o.rb = jb.rb; o.tlen = jb.tlen; o.state = n_b <= bcap ? 1 : 0; o.score = gmax; o.te = te; o.qe = qe; o.score2 = -1; o.te2 = -1; o.tb = o.qb = -1; o._pad = 0;
			if (n_b && n_b <= bcap) {
				const int k = (gmax + maxsc - 1) / maxsc, low = te - k, high = te + k;
				for (int x = 0; x < n_b; ++x) {
					const unsigned long long v = bl[x]; const int e = (int)(uint32_t)v, sc = (int)(v >> 32);
					if ((e < low || e > high) && sc > o.score2) { o.score2 = sc; o.te2 = e; }
				}
			}
			res[slot] = o;
			ncell += (unsigned long long)jb.tlen * jb.qlen;
		}
		}
	}
	if (cells && ncell) atomicAdd(cells, ncell);
}

/* the jobs whose forward pass calls for the reverse pass (upstream ksw_align2: KSW_XSTART, and with KSW_XSUBO only a score that reached minsc):
 * one lane per sorted job; keys[] = padded length of the reversed query prefix << 48 | rows (te + 1) << 32 | slot; n_out[1] = most rows */
__global__ void __launch_bounds__(64) ssg_k_msw_revlist(long n_jobs, const uint64_t *sorted, const ssg_msjob_t *jobs, const ssg_msres_t *res, long n_slots, uint64_t *keys, unsigned int *n_out)
{
	const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_jobs) return;
	const long slot = (long)(uint32_t)sorted[t];
	if (slot >= n_slots) return;
	const ssg_msjob_t jb = jobs[slot]; const ssg_msres_t r = res[slot];
	if (r.state != 1 || !jb.xstart || (jb.minsc != 0x10000 && r.score < jb.minsc) || r.te < 0 || r.qe < 0 || (jb.p != 8 && jb.p != 16)) return;
	const int qp = (r.qe + 1 + jb.p - 1) / jb.p * jb.p;
	keys[atomicAdd(n_out, 1u)] = (uint64_t)qp << 48 | (uint64_t)(r.te + 1) << 32 | (uint64_t)slot;
	atomicMax(n_out + 1, (unsigned int)(r.te + 1));
}

/* ---------------- which windows will mem_matesw align?  (the lists before any rescue) ---------------- */

/* one lane per listed pair k: cnt[2k + i] = 4 * (anchors of side i): upstream's b[i] = the first max_matesw hits within pen_unpaired of the best */
__global__ void __launch_bounds__(64) ssg_k_msw_count(ssg_mem_opt_t opt, int n_todo, const int32_t *todo, const int64_t *reg_off, const ssg_alnreg_t *regs, const int32_t *n_reg, int32_t *cnt)
{
	const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_todo) return;
	const long p = todo[k];
	for (int i = 0; i < 2; ++i) {
		const ssg_alnreg_t *a = regs + reg_off[2*p + i];
		const int an = n_reg[2*p + i], thr = an ? a[0].score - opt.pen_unpaired : 0;
		int c = 0;
		for (int j = 0; j < an && c < opt.max_matesw && c < 64; ++j) if (a[j].score >= thr) ++c;
		cnt[2*k + i] = 4 * c;
	}
}

/* one lane per anchor g (slot base[...] / 4 numbering): mem_matesw's test `is there already a hit inside the window of this orientation'
 * on the lists as they are now, then the window itself (upstream mem_matesw's rb / re, bns_fetch_seq's clipping to the contig at the
 * window's middle) -> jobs[slot], keys[atomic position] = qp << 48 | tlen << 32 | slot. */
__global__ void __launch_bounds__(64) ssg_k_msw_emit(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_todo, long n_anchor, const int32_t *todo, const int64_t *base,
                               const int64_t *read_off, const int64_t *reg_off, const ssg_alnreg_t *regs, const int32_t *n_reg,
                               const int32_t *pair_batch, const ssg_pestat_t *pes_all, ssg_msjob_t *jobs, uint64_t *keys, unsigned int *n_jobs /* [2]: windows listed, the longest of them */)
{
	const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_anchor) return;
	int lo = 0, hi = 2 * n_todo - 1;   /* last side with base[side] <= 4g */
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (base[mid] <= 4 * g) lo = mid; else hi = mid - 1; }
	const int side = lo, i = side & 1, j = (int)(g - (base[side] >> 2));
	const long p = todo[side >> 1];
	const ssg_pestat_t *pes = pes_all + (long)pair_batch[p] * 4;
	const ssg_alnreg_t *a = regs + reg_off[2*p + i], *ma = regs + reg_off[2*p + !i];
	const int an = n_reg[2*p + i], man = n_reg[2*p + !i];
	const int thr = an ? a[0].score - opt.pen_unpaired : 0;
	int at = -1;
	for (int x = 0, c = 0; x < an; ++x) if (a[x].score >= thr) { if (c == j) { at = x; break; } ++c; }
	if (at < 0) return;
	const int64_t arb = a[at].rb, l_pac = ix.l_pac; const int arid = a[at].rid;
	const int l_ms = (int)(read_off[2*p + !i + 1] - read_off[2*p + !i]);
	if (!ssg_ml_fits(opt, l_ms)) return;
	int skip[4];
	for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (int m = 0; m < man; ++m) {
		int64_t dist;
		const int r = ssg_infer_dir(l_pac, arb, ma[m].rb, &dist);
		if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
	}
	for (int r = 0; r < 4; ++r) {
		if (skip[r]) continue;
		const int is_rev = (r >> 1 != (r & 1)), is_larger = !(r >> 1);
		int64_t rb, re; int rid = -1;
		if (!is_rev) {
			rb = is_larger ? arb + pes[r].low : arb - pes[r].high;
			re = (is_larger ? arb + pes[r].high : arb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? arb + pes[r].low : arb - pes[r].high) - l_ms;
			re = is_larger ? arb + pes[r].high : arb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) {
			int rv; rid = ssg_pos2rid(ix, ssg_depos(ix, (rb + re) >> 1, &rv));
			int64_t far_beg = ix.ctg_off[rid], far_end = far_beg + ix.ctg_len[rid];
			if (rv) { const int64_t t2 = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t2; }
			rb = rb > far_beg ? rb : far_beg;
			re = re < far_end ? re : far_end;
		}
		if (!(arid == rid && re - rb >= opt.min_seed_len)) continue;
		if (re - rb > SSG_ML_TMAX) continue;
		const int xtra = SSG_KSW_XSUBO | SSG_KSW_XSTART | (l_ms * opt.a < 250 ? SSG_KSW_XBYTE : 0) | (opt.min_seed_len * opt.a);
		ssg_msjob_t jb; jb.rb = rb; jb.qoff = read_off[2*p + !i]; jb.tlen = (int)(re - rb); jb.qlen = l_ms; jb.qp = ssg_align2_qp(l_ms, xtra);
		jb.minsc = xtra & 0xffff; jb.is_rev = is_rev; jb.p = (xtra & SSG_KSW_XBYTE) ? 16 : 8; jb.xstart = 1; jb._pad = 0;
		const long slot = base[side] + 4 * j + r;
		jobs[slot] = jb;
		keys[atomicAdd(n_jobs, 1u)] = (uint64_t)jb.qp << 48 | (uint64_t)jb.tlen << 32 | (uint64_t)slot;
		atomicMax(n_jobs + 1, (unsigned int)jb.tlen);
	}
}
#endif
