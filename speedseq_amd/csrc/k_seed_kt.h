/*
 * k_seed_kt.h -- the seeding kernel with a table of short-pattern intervals, and the kernels that build / check that table
 * (opt-in: SSG_KTAB_K).  Compiled in a translation unit of its own (ssg_ktab.cpp): on ROCm 7.2 the machine
 * code hipcc makes of the product's seeding kernel changed -- and its results on the MI355X went wrong, emulation unaffected -- as soon
 * as these kernels shared a translation unit with it (GPU bisect of round 3), so nothing here is visible to
 * ssgpu_core.cpp's device code.  The small helpers below are copies of k_seed.h's.
 */
#ifndef SSG_K_SEED_KT_H
#define SSG_K_SEED_KT_H
#include "ssg_dev.h"

SSG_DEVFN void ssg_set_intv(const ssg_index_view_t &ix, int c, ssg_intv_t &ik)
{
	ik.x0 = ix.L2[c] + 1; ik.x2 = ix.L2[c+1] - ix.L2[c]; ik.x1 = ix.L2[3-c] + 1; ik.info = 0;
}

#define SSG_SM_QWORDS 32   /* 8 bases per word: reads up to 256 bases */
enum { SM_FWD = 0, SM_BWD, SM_P3F, SM_READ, SM_P1, SM_P2, SM_P3, SM_OUT, SM_FIN };   /* the three hot states first: the compiler lowers the dispatch to a comparison tree over the value */
enum { SM_PEND_NONE = 0, SM_PEND_FWD, SM_PEND_BWD, SM_PEND_P3 };

/* interval-list entries in scratch and in the carried registers: 16 bytes (x0, x1, x2 < 2^40; info = end position < 256) */
struct alignas(16) ssg_pk_t { uint64_t w0, w1; };
SSG_DEVFN ssg_pk_t ssg_pk(const ssg_intv_t &v)
{ ssg_pk_t p; p.w0 = v.x0 | (v.x1 & 0xffffffull) << 40; p.w1 = (v.x1 >> 24) | v.x2 << 16 | v.info << 56; return p; }
SSG_DEVFN ssg_intv_t ssg_unpk(const ssg_pk_t &p)
{ ssg_intv_t v; v.x0 = p.w0 & 0xffffffffffull; v.x1 = (p.w0 >> 40) | (p.w1 & 0xffffull) << 24; v.x2 = (p.w1 >> 16) & 0xffffffffffull; v.info = p.w1 >> 56; return v; }

/* ---- table of short-pattern intervals (ssg_index.ktab) ---- */
SSG_DEVFN long ssg_ktab_off(int j) { return (long)(((1ull << (2 * j)) - 4ull) / 3ull); }   /* entries of the levels below j */
/* level j from level j - 1: the four one-base left extensions of every pattern, by upstream's own bwt_extend (is_back = 1), so an
 * entry is bit for bit what the extension it stands in for returns (the interval of a pattern does not depend on the order in
 * which it was extended to).  Pattern code: little-endian base 4 (first base least significant): children of p are 4p .. 4p + 3. */
__global__ void ssg_k_ktab_level(ssg_index_view_t ix, int j, ssg_pk_t *tab)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= (1L << (2 * (j - 1)))) return;
	ssg_pk_t *const out = tab + ssg_ktab_off(j) + 4 * p;
	ssg_intv_t ok[4];
	if (j == 1) { for (int c = 0; c < 4; ++c) ssg_set_intv(ix, c, ok[c]); }
	else ssg_bwt_extend(ix, ssg_unpk(tab[ssg_ktab_off(j - 1) + p]), ok, 1);
	for (int c = 0; c < 4; ++c) { ok[c].info = 0; out[c] = ssg_pk(ok[c]); }
}
/* the same level from level j - 1 by one-base RIGHT extensions with the seeding kernel's own ssg_bwt_extend1_lean (SSG_KTAB_BUILD=fwd):
 * pattern p of j - 1 bases followed by base c is entry p + c * 4^(j-1) */
__global__ void ssg_k_ktab_level_fwd(ssg_index_view_t ix, int j, ssg_pk_t *tab)
{
	const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, np = 1L << (2 * (j - 1));
	if (t >= 4 * np) return;
	const long p = t % np; const int c = (int)(t / np);
	ssg_intv_t o;
	if (j == 1) ssg_set_intv(ix, c, o);
	else o = ssg_bwt_extend1_lean(ix, ssg_unpk(tab[ssg_ktab_off(j - 1) + p]), 3 - c, 0);
	o.info = 0;
	tab[ssg_ktab_off(j) + p + (long)c * np] = ssg_pk(o);
}
/* self-check of the table (SSG_KTAB_VERIFY=1): every `stride`-th pattern of level j once more, this time the way the seeding kernel
 * would have reached it without the table -- from its first base by forward extensions (ssg_bwt_extend1_lean) -- and compared */
__global__ void ssg_k_ktab_verify(ssg_index_view_t ix, int j, long stride, const ssg_pk_t *tab, unsigned long long *bad)
{
	const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, code = t * stride;
	if (code >= (1L << (2 * j))) return;
	ssg_intv_t ik;
	ssg_set_intv(ix, (int)(code & 3), ik);
	for (int k = 1; k < j; ++k) ik = ssg_bwt_extend1_lean(ix, ik, 3 - (int)((code >> (2 * k)) & 3), 0);
	const ssg_intv_t e = ssg_unpk(tab[ssg_ktab_off(j) + code]);
	if (ik.x2 != e.x2 || (ik.x2 && (ik.x0 != e.x0 || ik.x1 != e.x1))) atomicAdd(&bad[j], 1ull);
}
/* code of the n <= 15 bases from position b of a read held as 4-bit codes, 8 per LDS word (word w of the read at qw[w * stride]); no base of the window is ambiguous */
SSG_DEVFN uint64_t ssg_smq_window(const uint32_t *qw, int stride, int b)
{	/* the 16 codes from position b, 4 bits each, first base lowest */
	const int w0 = b >> 3, sh = (b & 7) << 2;
	const int w1 = w0 + 1 < SSG_SM_QWORDS ? w0 + 1 : SSG_SM_QWORDS - 1, w2 = w0 + 2 < SSG_SM_QWORDS ? w0 + 2 : SSG_SM_QWORDS - 1;
	uint64_t v = ((uint64_t)qw[w0 * stride] | (uint64_t)qw[w1 * stride] << 32) >> sh;
	if (sh) v |= (uint64_t)qw[w2 * stride] << (64 - sh);
	return v;
}
SSG_DEVFN uint32_t ssg_smq_code(const uint32_t *qw, int stride, int b, int n)
{
	uint64_t v = ssg_smq_window(qw, stride, b);
	v &= 0x3333333333333333ull; v = (v | v >> 2) & 0x0f0f0f0f0f0f0f0full; v = (v | v >> 4) & 0x00ff00ff00ff00ffull;
	v = (v | v >> 8) & 0x0000ffff0000ffffull; v = (v | v >> 16) & 0xffffffffull;
	return (uint32_t)v & ((1u << (2 * n)) - 1u);
}

#ifndef SSG_SMQ_WAVES
#define SSG_SMQ_WAVES 4
#endif
#ifndef SSG_SMQ_TRIPS
#define SSG_SMQ_TRIPS 2
#endif
/* diagnostic builds only (`make ktvariant`, tools/dbg/kt_variants.sh): counters / kernarg echo read back by ssg_ktab_dbg_read */
#ifdef SSG_KT_DBG
#ifdef SSG_EMU
static unsigned long long ssg_kt_dbg[64];
#else
__device__ unsigned long long ssg_kt_dbg[64];
#endif
#ifdef SSG_EMU
#define KTD(i, v) (ssg_kt_dbg[i] += (unsigned long long)(v))
#else
#define KTD(i, v) atomicAdd(&ssg_kt_dbg[i], (unsigned long long)(v))
#endif
#else
#define KTD(i, v) ((void)0)
#endif
#ifdef SSG_KT_UNIFORM   /* = all three: wave-uniform loop exit, the site as one predicated block, no `continue' out of the middle of the site */
#define SSG_KT_U_EXIT
#define SSG_KT_U_SITE
#define SSG_KT_U_INNER
#endif
#ifdef SSG_KT_NOTAB
#define SSG_KT_ON false
#else
#define SSG_KT_ON true
#endif
/* ssg_k_smem_quad with the table of short-pattern intervals (kt_tab, kt_k): an extension whose result pattern has at most kt_k bases is one
 * 16-byte load, and the third pass starts kt_k bases in.  A kernel of its own, so that the default kernel above keeps the exact text (and
 * machine code) of rounds 1-2; opt-in with SSG_KTAB_K until its GPU results match the oracle. */
template <int LPR>
__global__ void __launch_bounds__(64, SSG_SMQ_WAVES) ssg_k_smem_quad_kt(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const int32_t *read_ids,
                           const uint8_t *seq, const int64_t *off,
                           ssg_intv_t *out_intv, int32_t *out_n, int cap,
                           ssg_intv_t *scratch, int scap, unsigned long long *n_extend, unsigned int *next_read, const ssg_pk_t *kt_tab, int kt_k)
{
	constexpr bool KT = SSG_KT_ON;
	constexpr int RPW = 64 / LPR;   /* reads per wave */
	__shared__ uint32_t qlds[SSG_SM_QWORDS * RPW];
	const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, nq = ((long)gridDim.x * blockDim.x) / LPR;
	const int lane = (int)(threadIdx.x & 63), Q = lane / LPR, ql = lane % LPR;
	/* per-wave slab of 2 lists x scap entries x 16 quads, entry e of quad Q at [e*16 + Q] */
	ssg_pk_t *const vec0 = (ssg_pk_t*)scratch + (gt >> 6) * 2 * scap * RPW + Q, *const vec1 = vec0 + (long)scap * RPW;
	const uint32_t *const ql_ = qlds + Q;
#define SMQ(i) ((int)((ql_[((i) >> 3) * RPW] >> (((i) & 7) << 2)) & 15u))
#define SMV(v, e) ((v)[(long)(e) * RPW])
#ifdef SSG_EMU
#define QW 1          /* fibers of a quad are not in lock step: every lane stores the (identical) value it will read back */
#else
#define QW (ql == 0)  /* one lane of the quad stores */
#endif
	const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
#ifdef SSG_KT_DBG
	if (gt == 0) {   /* what the kernel sees of its by-value arguments */
		unsigned long long *d = ssg_kt_dbg + 16;
		d[0] = (unsigned long long)opt.min_seed_len; d[1] = (unsigned long long)opt.split_width; d[2] = opt.max_mem_intv; { uint32_t sfb; memcpy(&sfb, &opt.split_factor, 4); d[3] = sfb; }
		d[4] = (unsigned long long)split_len; d[5] = (unsigned long long)kt_k; d[6] = ix.primary; d[11] = ix.L2[4];
		d[12] = ix.seq_len; d[13] = (unsigned long long)scap; d[14] = (unsigned long long)cap; d[15] = (unsigned long long)n_reads; d[16] = (unsigned long long)(uintptr_t)kt_tab; d[17] = (unsigned long long)(uintptr_t)ix.bwt;
		d[18] = (unsigned long long)gridDim.x; d[19] = (unsigned long long)blockDim.x; d[20] = (unsigned long long)opt.max_occ; d[21] = (unsigned long long)ix.sa_intv;
	}
#endif
	unsigned long long my_nx = 0;
	long it = gt / LPR - nq;
	int state = SM_READ, pend = SM_PEND_NONE;
	int len = 0, x = 0, k = 0, old_n = 0, caller = 0, mem_n = 0, ovf = 0;
	ssg_intv_t *mem = 0;
	int sx = 0, i = 0, j = 0, curr_n = 0, prev_n = 0, prev_rev = 0, flip = 0, m1_n = 0, m1_last_beg = 0, ret = 0, e_c = 0;
	uint64_t min_intv = 1, last_x2 = 0;
	ssg_intv_t ik, p;
	ik.x0 = ik.x1 = ik.x2 = ik.info = 0; p = ik;
	ssg_pk_t pn, c0, first; pn.w0 = pn.w1 = 0; c0 = first = pn;
	/* transitions done where they arise instead of through a state of their own (one dispatch less on the way):
	 * the forward list becomes `prev', walked from its top (= ik), ret = end of the longest match; return of bwt_smem1a to its caller */
#define SM_DO_FWDEND() do { ret = (int)ik.info; flip ^= 1; prev_n = curr_n < scap ? curr_n : scap; prev_rev = 1; curr_n = 0; i = sx - 1; j = 0; first = ssg_pk(ik); state = SM_BWD; } while (0)
/* next start position of the third pass (upstream bwt_seed_strategy1 from every position): skip ambiguous bases, open the interval */
/* With the table (KT) the first kt_k - 1 extensions of a start collapse into one look-up: upstream records nothing before the pattern has
 * min_seed_len (> kt_k) bases, so only an ambiguous base or the read's end inside the window matters -- the start then moves past it
 * exactly as upstream's loop returns, one window per trip (state SM_P3 comes back here); the skipped bwt_extend calls still count as
 * algorithmic work. */
#define SM_DO_P3() do { \
		while (x < len && SMQ(x) > 3) ++x; \
		if (x >= len) state = SM_OUT; \
		else if (!KT || kt_k < 2 || kt_k >= opt.min_seed_len) { ssg_set_intv(ix, SMQ(x), ik); i = x + 1; state = SM_P3F; } \
		else { \
			const int kk_ = kt_k; \
			const unsigned long long nm_ = ssg_smq_window(ql_, RPW, x) & 0x4444444444444444ull; \
			int run_ = nm_ ? (int)((__ffsll((unsigned long long)nm_) - 1) >> 2) : 16; \
			run_ = run_ > len - x ? len - x : run_; \
			if (run_ >= kk_) { ik = ssg_unpk(kt_tab[ssg_ktab_off(kk_) + (long)ssg_smq_code(ql_, RPW, x, kk_)]); i = x + kk_; my_nx += (unsigned long long)(kk_ - 1); state = SM_P3F; } \
			else { my_nx += (unsigned long long)(run_ - 1); if (x + run_ >= len) { x = len; state = SM_OUT; } else { x += run_ + 1; state = SM_P3; } } \
		} \
	} while (0)
#define SM_DO_RET() do { if (caller == 1) { x = ret; state = SM_P1; } else { ++k; state = SM_P2; } } while (0)
	unsigned long long tn_adv = 0, tn_ext = 0, tn_rounds = 0, tn_ready = 0, tn_alive = 0, tn_t0 = 0;   /* SSG_TUNING only: cycles in the state machine / at the extension site, rounds, ready and live lanes per round */
	for (;;) {
		if (SSG_TUNING) tn_t0 = ssg_clock();
		/* a bounded number of state-machine steps per extension round: a lane in the middle of a transition sits the round out instead of
		 * making the whole wave spin through the switch again (the wave pays for every trip, whoever needs it) */
		SSG_UNROLL for (int trip = 0; trip < SSG_SMQ_TRIPS; ++trip) if (pend == SM_PEND_NONE && state != SM_FIN) {
			ssg_pk_t *const curr = flip ? vec1 : vec0;
			/* the three states a lane is in nearly all the time (forward loop, backward loop, third pass) first, straight-line; a wave whose
			 * lanes are all there skips the switch over the rare states with one branch */
			if (state == SM_FWD) { /* top of upstream's forward loop: for (i = x + 1; i < len; ++i) */
				if (i < len && SMQ(i) < 4) { pend = SM_PEND_FWD; e_c = 3 - SMQ(i); }
				else { if (curr_n < scap) { if (QW) SMV(curr, curr_n) = ssg_pk(ik); } else ovf = 1; ++curr_n; SM_DO_FWDEND(); }
			} else if (state == SM_BWD) { /* for (i = x - 1; i >= -1; --i) for (j = 0; j < prev->n; ++j) */
				if (j >= prev_n) {
					if (curr_n == 0) SM_DO_RET();
					else {
						flip ^= 1; prev_n = curr_n < scap ? curr_n : scap; prev_rev = 0; curr_n = 0; j = 0; --i; first = c0;
						if (i < -1) SM_DO_RET();
					}
				} else {
					p = ssg_unpk(j == 0 ? first : pn);
					const int cb = i < 0 ? -1 : SMQ(i) < 4 ? SMQ(i) : -1;
					if (cb >= 0) { pend = SM_PEND_BWD; e_c = cb; }
					else { /* no base to extend with: only the first (longest) interval of the row can be an SMEM, the rest are no-ops */
						if (j == 0 && (m1_n == 0 || i + 1 < m1_last_beg)) {
							++m1_n; m1_last_beg = i + 1;
							if ((int)(uint32_t)p.info - (i + 1) >= opt.min_seed_len) {
								ssg_intv_t o = p; o.info |= (uint64_t)(i + 1) << 32;
								if (mem_n < cap) { if (QW) mem[mem_n] = o; } else ovf = 1;
								++mem_n;
							}
						}
						j = prev_n;
					}
				}
			} else if (state == SM_P3F) {
				if (i < len) {
					if (SMQ(i) < 4) { pend = SM_PEND_P3; e_c = 3 - SMQ(i); }
					else { x = i + 1; SM_DO_P3(); }
				} else { x = len; state = SM_OUT; }
			} else
			switch (state) {
			case SM_READ: {
				/* next read: a shared counter when every lane owns a read (evens out the per-read cost), a fixed stride for quads */
				if (LPR == 1 && next_read) it = (long)atomicAdd(next_read, 1u); else it += nq;
				if (it >= n_reads) { state = SM_FIN; break; }
				const int r = read_ids ? read_ids[it] : (int)it;
				const uint8_t *q = seq + off[r];
				len = (int)(off[r+1] - off[r]);
				mem = out_intv + (long)it * cap; mem_n = 0; ovf = 0;
				{	/* the read as 4-bit codes, 8 per LDS word (all lanes of a quad write the same words: no cross-lane hand-off needed).
					 * Fetched as aligned 8-byte words, four LDS words per round trip: this runs with one lane of the wave active. */
					const unsigned al = (unsigned)((uintptr_t)q & 7), sh8 = al << 3;
					const uint64_t *const qa = (const uint64_t*)(q - al);
					const int nw = (len + 7) >> 3, nb = (int)al + len;   /* nb: bytes from qa to the read's end */
					for (int w0 = 0; w0 < nw; w0 += 4) {
						uint64_t t[5];
						SSG_UNROLL for (int jj = 0; jj < 5; ++jj) t[jj] = (w0 + jj) * 8 < nb ? qa[w0 + jj] : 0;
						SSG_UNROLL for (int jj = 0; jj < 4; ++jj) {
							const int w = w0 + jj;
							if (w >= nw) break;
							uint64_t v = sh8 ? (t[jj] >> sh8) | (t[jj + 1] << (64 - sh8)) : t[jj];
							v &= 0x0f0f0f0f0f0f0f0full;
							v = (v | v >> 4) & 0x00ff00ff00ff00ffull;
							v = (v | v >> 8) & 0x0000ffff0000ffffull;
							uint32_t v32 = (uint32_t)(v | v >> 16);
							const int nv = len - w * 8;
							if (nv < 8) v32 &= (1u << (nv << 2)) - 1u;
							qlds[w * RPW + Q] = v32;
						}
					}
				}
				x = 0;
				state = len >= opt.min_seed_len ? SM_P1 : SM_OUT;
			} break;
			case SM_P1:
				if (x >= len) { old_n = mem_n < cap ? mem_n : cap; k = 0; state = SM_P2; }
				else if (SMQ(x) > 3) ++x;
				else { sx = x; min_intv = 1; caller = 1; state = SM_FWD; m1_n = 0; curr_n = 0; i = sx + 1;
				       ssg_set_intv(ix, SMQ(sx), ik); ik.info = (uint64_t)(sx + 1); }
				break;
			case SM_P2: /* re-seed from the middle of long SMEMs with few occurrences */
				if (k >= old_n) { x = 0; state = opt.max_mem_intv > 0 ? SM_P3 : SM_OUT; break; }
				{
					const ssg_intv_t m = mem[k];
					const int start = (int)(m.info >> 32), end = (int)(uint32_t)m.info;
					if (end - start < split_len || m.x2 > (uint64_t)opt.split_width) { ++k; break; }
					sx = (start + end) >> 1; min_intv = m.x2 + 1; caller = 2; m1_n = 0; curr_n = 0; i = sx + 1;
					if (SMQ(sx) > 3) { SM_DO_RET(); break; }   /* bwt_smem1a returns at once on an ambiguous base */
					ssg_set_intv(ix, SMQ(sx), ik); ik.info = (uint64_t)(sx + 1);
					state = SM_FWD;
				}
				break;
			case SM_P3:
				SM_DO_P3();
				break;
			case SM_OUT:
				if (QW) out_n[it] = ovf ? -1 : mem_n;
				state = SM_READ;
				break;
			}
		}
		if (SSG_TUNING) { const unsigned long long t1 = ssg_clock(); tn_adv += t1 - tn_t0; tn_t0 = t1; ++tn_rounds; tn_ready += (unsigned long long)__popcll(wv_ballot(pend != SM_PEND_NONE)); tn_alive += (unsigned long long)__popcll(wv_ballot(state != SM_FIN)); }
#ifdef SSG_KT_U_EXIT   /* wave-uniform loop exit */
		if (!wv_ballot(state != SM_FIN)) break;
#else
		if (state == SM_FIN) break;
#endif
#ifdef SSG_KT_U_SITE   /* the extension site as one predicated block */
		if (pend != SM_PEND_NONE) {
#else
		if (pend == SM_PEND_NONE) continue;
#endif
		KTD(0, 1); if (pend == SM_PEND_NONE) KTD(1, 1); if (state == SM_FIN) KTD(2, 1); if (pend < 0 || pend > SM_PEND_P3) KTD(3, 1);
		/* ---- the one extension site: two rank-block quarters per lane + the next list entry ---- */
		ssg_wave_ldssync();   /* list entries stored by lane 0 of the quad last iteration are read by all four below (same wave: in order on the GPU) */
		const ssg_pk_t *const prev = flip ? vec0 : vec1;
		const int back = pend == SM_PEND_BWD;
		const int jn = back && j + 1 < prev_n ? j + 1 : 0;
		ssg_pk_t pf; pf.w0 = pf.w1 = 0;
		if (jn) pf = SMV(prev, prev_rev ? prev_n - 1 - jn : jn);   /* issued together with the rank-block loads below */
		/* the pattern the extension ends with: [i, end of p) going left, [start, i] going right; up to kt_k bases its interval is in the table */
		const int pat_b = back ? i : pend == SM_PEND_FWD ? sx : x, pat_n = (back ? (int)p.info : i + 1) - pat_b;
		ssg_intv_t okc;
#ifdef SSG_KT_CHECK   /* both ways; the extension's result is used, a differing table entry counted */
		okc = LPR == 4 ? ssg_bwt_extend1_quad(ix, back ? p : ik, e_c, back, ql) : ssg_bwt_extend1_lean(ix, back ? p : ik, e_c, back);
		if (KT && pat_n <= kt_k) {
			const ssg_intv_t te = ssg_unpk(kt_tab[ssg_ktab_off(pat_n) + (long)ssg_smq_code(ql_, RPW, pat_b, pat_n)]);
			KTD(4, 1);
			if (te.x2 != okc.x2 || (te.x2 && (te.x0 != okc.x0 || te.x1 != okc.x1))) { KTD(5, 1); if (back) KTD(6, 1); else if (pend == SM_PEND_FWD) KTD(7, 1); else KTD(8, 1); }
		}
#else
		if (KT && pat_n <= kt_k) { okc = ssg_unpk(kt_tab[ssg_ktab_off(pat_n) + (long)ssg_smq_code(ql_, RPW, pat_b, pat_n)]); KTD(4, 1); }
		else okc = LPR == 4 ? ssg_bwt_extend1_quad(ix, back ? p : ik, e_c, back, ql) : ssg_bwt_extend1_lean(ix, back ? p : ik, e_c, back);
#endif
		++my_nx;
		{
			ssg_pk_t *const curr = flip ? vec1 : vec0;
			if (pend == SM_PEND_FWD) {
#ifdef SSG_KT_U_INNER
				bool fwd_end = false;
				if (okc.x2 != ik.x2) {
					if (curr_n < scap) { if (QW) SMV(curr, curr_n) = ssg_pk(ik); } else ovf = 1;
					++curr_n;
					if (okc.x2 < min_intv) { SM_DO_FWDEND(); fwd_end = true; }   /* break: ik stays the last pushed */
				}
				if (!fwd_end) { ik = okc; ik.info = (uint64_t)(i + 1); ++i; }
#else
				if (okc.x2 != ik.x2) {
					if (curr_n < scap) { if (QW) SMV(curr, curr_n) = ssg_pk(ik); } else ovf = 1;
					++curr_n;
					if (okc.x2 < min_intv) { SM_DO_FWDEND(); pend = SM_PEND_NONE; continue; }   /* break: ik stays the last pushed */
				}
				ik = okc; ik.info = (uint64_t)(i + 1); ++i;
#endif
			} else if (back) {
				pn = pf;
				if (okc.x2 < min_intv) {
					if (curr_n == 0 && (m1_n == 0 || i + 1 < m1_last_beg)) {
						++m1_n; m1_last_beg = i + 1;
						if ((int)(uint32_t)p.info - (i + 1) >= opt.min_seed_len) {
							ssg_intv_t o = p; o.info |= (uint64_t)(i + 1) << 32;
							if (mem_n < cap) { if (QW) mem[mem_n] = o; } else ovf = 1;
							++mem_n;
						}
					}
				} else if (curr_n == 0 || okc.x2 != last_x2) {
					ssg_intv_t o = okc; o.info = p.info;
					const ssg_pk_t po = ssg_pk(o);
					if (curr_n == 0) c0 = po;
					if (curr_n < scap) { if (QW) SMV(curr, curr_n) = po; } else ovf = 1;
					++curr_n; last_x2 = okc.x2;
				}
				++j;
			} else { /* SM_PEND_P3 */
				if (okc.x2 < (uint64_t)opt.max_mem_intv && i - x >= opt.min_seed_len) {
					if (okc.x2 > 0) {
						ssg_intv_t o = okc; o.info = (uint64_t)x << 32 | (uint64_t)(i + 1);
						if (mem_n < cap) { if (QW) mem[mem_n] = o; } else ovf = 1;
						++mem_n;
					}
					x = i + 1; SM_DO_P3();
				} else { ik = okc; ++i; }
			}
			pend = SM_PEND_NONE;
		}
#ifdef SSG_KT_U_SITE
		}
#endif
		if (SSG_TUNING) tn_ext += ssg_clock() - tn_t0;
	}
	if (SSG_TUNING) {   /* slots 24..28: the lane that ran longest speaks for its wave (all live lanes of a wave count the same rounds) */
		const int a = wv_max((int)(tn_adv >> 4)), e = wv_max((int)(tn_ext >> 4)), r = wv_max((int)tn_rounds), y = wv_max((int)(tn_ready >> 6)), v = wv_max((int)(tn_alive >> 6));
		if (lane == 0) { atomicAdd(&ssg_dbg_cyc[24], (unsigned long long)a << 4); atomicAdd(&ssg_dbg_cyc[25], (unsigned long long)e << 4); atomicAdd(&ssg_dbg_cyc[26], (unsigned long long)r); atomicAdd(&ssg_dbg_cyc[27], (unsigned long long)y); atomicAdd(&ssg_dbg_cyc[28], (unsigned long long)v); }
	}
#undef SMQ
#undef SMV
#undef QW
#undef SM_DO_FWDEND
#undef SM_DO_RET
#undef SM_DO_P3
	if (n_extend && my_nx && ql == 0) atomicAdd(n_extend, my_nx);
}
/* self-check of the denser table (SSG_SA_VERIFY=1): every `stride`-th entry against upstream's own bwt_sa walk on the file's samples */
__global__ void ssg_k_sa_verify(ssg_index_view_t ix, int new_intv, const uint64_t *sa_new, long n_new, long stride, unsigned long long *bad)
{
	const long j = ((long)blockIdx.x * blockDim.x + threadIdx.x) * stride;
	if (j >= n_new) return;
	const uint64_t r = (uint64_t)j * (uint64_t)new_intv;
	const uint64_t want = (r % (uint64_t)ix.sa_intv) == 0 ? ix.sa[r / (uint64_t)ix.sa_intv] : ssg_bwt_sa(ix, r);
	if (sa_new[j] != want) atomicAdd(bad, 1ull);
}
#endif
