/*
 * k_seed.h -- gfx950 kernels for FM-index seeding.
 *
 *   ssg_k_smem_quad one lane per read (or four: SSG_SMEM_LPR=4, quad-cooperative rank-block fetch), state machine with one bwt_extend site (the product path);
 *   ssg_k_smem_lane one lane per read, nested loops as upstream writes them (SSG_SMEM_KERNEL=lane: A/B runs);
 *                   both: the three SMEM passes of upstream mem_collect_intv (bwt_smem1a x2 +
 *                   bwt_seed_strategy1), intervals sorted by (start,end).  Every bwt_extend is two rank
 *                   queries = two random 64-byte lines.
 *   ssg_k_sal_count per interval: number of sampled occurrences (<= max_occ) -> prefix sum.
 *   ssg_k_sal       one lane per (interval, occurrence): upstream bwt_sa LF-walk + sampled-SA
 *                   gather, then bns_intv2rid; writes mem_seed_t in upstream visiting order.
 */
#ifndef SSG_K_SEED_H
#define SSG_K_SEED_H
#include "ssg_dev.h"

/* interval vector; `st` = element stride: the per-lane work vectors are interleaved across the 64
 * lanes of a wave (entry e of lane l at base[e*64 + l]) so that lanes pushing / reading their e-th
 * entry together touch one contiguous 2-KB span instead of 64 scattered half-used lines */
struct ssg_ivec_t { ssg_intv_t *a; int n, cap; int ovf; unsigned nx; /* bwt_extend calls (rank-query accounting) */ int st; };
#define IV(v, i) ((v).a[(long)(i) * (v).st])
SSG_DEVFN void iv_push(ssg_ivec_t &v, const ssg_intv_t &x) { if (v.n < v.cap) IV(v, v.n) = x; else v.ovf = 1; ++v.n; }
SSG_DEVFN void iv_reverse(ssg_ivec_t &v)
{
	int n = v.n < v.cap ? v.n : v.cap;
	for (int j = 0; j < n >> 1; ++j) { ssg_intv_t t = IV(v, n-1-j); IV(v, n-1-j) = IV(v, j); IV(v, j) = t; }
}
SSG_DEVFN void ssg_set_intv(const ssg_index_view_t &ix, int c, ssg_intv_t &ik)
{
	ik.x0 = ix.L2[c] + 1; ik.x2 = ix.L2[c+1] - ix.L2[c]; ik.x1 = ix.L2[3-c] + 1; ik.info = 0;
}

/* upstream bwt_smem1a (max_intv == 0 as in mem_collect_intv) */
SSG_DEVFN int ssg_smem1(const ssg_index_view_t &ix, int len, const uint8_t *q, int x, uint64_t min_intv,
                        ssg_ivec_t &mem, ssg_ivec_t &va, ssg_ivec_t &vb)
{
	int i, j, c, ret;
	ssg_intv_t ik;
	ssg_ivec_t *prev = &va, *curr = &vb, *swap;
	mem.n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	ssg_set_intv(ix, q[x], ik);
	ik.info = (uint64_t)(x + 1);
	for (i = x + 1, curr->n = 0; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			const ssg_intv_t okc = ssg_bwt_extend1_lean(ix, ik, c, 0); ++mem.nx;
			if (okc.x2 != ik.x2) {
				iv_push(*curr, ik);
				if (okc.x2 < min_intv) break;
			}
			ik = okc; ik.info = (uint64_t)(i + 1);
		} else { iv_push(*curr, ik); break; }
	}
	if (i == len) iv_push(*curr, ik);
	iv_reverse(*curr);
	ret = (int)IV(*curr, 0).info;
	swap = curr; curr = prev; prev = swap;
	for (i = x - 1; i >= -1; --i) {
		c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		for (j = 0, curr->n = 0; j < prev->n; ++j) {
			ssg_intv_t p = IV(*prev, j);
			ssg_intv_t okc; okc.x0 = okc.x1 = okc.x2 = okc.info = 0;
			if (c >= 0) { okc = ssg_bwt_extend1_lean(ix, p, c, 1); ++mem.nx; }
			if (c < 0 || okc.x2 < min_intv) {
				if (curr->n == 0) {
					if (mem.n == 0 || (uint64_t)(i + 1) < (IV(mem, mem.n-1).info >> 32)) {
						ik = p; ik.info |= (uint64_t)(i + 1) << 32;
						iv_push(mem, ik);
					}
				}
			} else if (curr->n == 0 || okc.x2 != IV(*curr, curr->n-1).x2) {
				okc.info = p.info;
				iv_push(*curr, okc);
			}
		}
		if (curr->n == 0) break;
		swap = curr; curr = prev; prev = swap;
	}
	iv_reverse(mem);
	return ret;
}

/* upstream bwt_seed_strategy1 */
SSG_DEVFN int ssg_seed_strategy1(const ssg_index_view_t &ix, int len, const uint8_t *q, int x, int min_len, uint64_t max_intv, ssg_intv_t &mem, unsigned &nx)
{
	int i, c;
	ssg_intv_t ik;
	mem.x0 = mem.x1 = mem.x2 = mem.info = 0;
	if (q[x] > 3) return x + 1;
	ssg_set_intv(ix, q[x], ik);
	for (i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			const ssg_intv_t okc = ssg_bwt_extend1_lean(ix, ik, c, 0); ++nx;
			if (okc.x2 < max_intv && i - x >= min_len) {
				mem = okc;
				mem.info = (uint64_t)x << 32 | (uint64_t)(i + 1);
				return i + 1;
			}
			ik = okc;
		} else return i + 1;
	}
	return len;
}

struct ssg_intv_lt { SSG_DEVMEM bool operator()(const ssg_intv_t &a, const ssg_intv_t &b) const { return a.info < b.info; } };

/*
 * One lane per read.  seq: concatenated nt4 codes; off[r]..off[r+1] delimit read r.
 * out_intv: [n_reads x cap] ; out_n: per-read interval count (count > cap => overflow, the host
 * re-runs those reads with a larger cap).  scratch: per launched lane 3*scap intervals.
 */
__global__ void __launch_bounds__(64) ssg_k_smem_lane(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const int32_t *read_ids,
                           const uint8_t *seq, const int64_t *off,
                           ssg_intv_t *out_intv, int32_t *out_n, int cap,
                           ssg_intv_t *scratch, int scap, unsigned long long *n_extend)
{
	long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
	ssg_intv_t *my = scratch + (gt >> 6) * 3 * scap * 64 + (gt & 63);   /* wave slab, lane-interleaved */
	unsigned long long my_nx = 0;
	for (long it = gt; it < n_reads; it += nt) {
		int r = read_ids ? read_ids[it] : (int)it;
		const uint8_t *q = seq + off[r];
		int len = (int)(off[r+1] - off[r]);
		ssg_ivec_t mem = { out_intv + (long)it * cap, 0, cap, 0, 0, 1 };
		ssg_ivec_t mem1 = { my, 0, scap, 0, 0, 64 }, va = { my + (long)scap * 64, 0, scap, 0, 0, 64 }, vb = { my + 2L * scap * 64, 0, scap, 0, 0, 64 };
		int x = 0, i, k, old_n;
		int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
		if (len >= opt.min_seed_len) {
			while (x < len) { /* pass 1 */
				if (q[x] < 4) {
					x = ssg_smem1(ix, len, q, x, 1, mem1, va, vb);
					for (i = 0; i < mem1.n; ++i) {
						ssg_intv_t p = IV(mem1, i);
						int slen = (int)((uint32_t)p.info - (uint32_t)(p.info >> 32));
						if (slen >= opt.min_seed_len) iv_push(mem, p);
					}
				} else ++x;
			}
			old_n = mem.n < mem.cap ? mem.n : mem.cap; /* pass 2 */
			for (k = 0; k < old_n; ++k) {
				ssg_intv_t p = mem.a[k];
				int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
				if (end - start < split_len || p.x2 > (uint64_t)opt.split_width) continue;
				ssg_smem1(ix, len, q, (start + end) >> 1, p.x2 + 1, mem1, va, vb);
				for (i = 0; i < mem1.n; ++i) {
					ssg_intv_t m = IV(mem1, i);
					if ((int)((uint32_t)m.info - (uint32_t)(m.info >> 32)) >= opt.min_seed_len) iv_push(mem, m);
				}
			}
			if (opt.max_mem_intv > 0) { /* pass 3 */
				x = 0;
				while (x < len) {
					if (q[x] < 4) {
						ssg_intv_t m;
						x = ssg_seed_strategy1(ix, len, q, x, opt.min_seed_len, opt.max_mem_intv, m, mem1.nx);
						if (m.x2 > 0) iv_push(mem, m);
					} else ++x;
				}
			}
			if (mem.n <= mem.cap && !mem1.ovf && !va.ovf && !vb.ovf) ssg_introsort(mem.a, (long)mem.n, ssg_intv_lt());
		}
		out_n[it] = (mem.ovf || mem1.ovf || va.ovf || vb.ovf) ? -1 : mem.n;
		my_nx += mem1.nx;
	}
	if (n_extend && my_nx) atomicAdd(n_extend, my_nx);
}

/*
 * ssg_k_smem_quad -- upstream mem_collect_intv with FOUR LANES PER READ and a single bwt_extend site.
 *
 * Shape of the problem: ~700 dependent bwt_extend per 150-bp read, each two random 64-byte rank blocks;
 * MI355X sustains ~55 G random lines/s (tools/dbg/gather_probe.cpp) and that, not 8 TB/s of streaming
 * bandwidth, is the roofline of this kernel.  Two things keep a straightforward one-lane-per-read kernel
 * (ssg_k_smem_lane below) at a fifth of it: the lanes of a wave sit in different loops of the nested
 * algorithm, so most extensions issue for a few lanes only, and every step is a chain of dependent round
 * trips (list entry -> extension -> list append).  Here
 *   - a quad owns a read: each lane fetches one 16-byte quarter of each rank block (one load instruction
 *     per block = 16 lines per wave; a per-lane fetch is 4 instructions x 64 lines) and the quad shares the
 *     popcounts by DPP; a wave runs 16 reads instead of 64, so divergence costs a quarter;
 *   - the three passes are a per-quad state machine (`advance': registers and LDS only) around ONE
 *     extension site per loop iteration, where every quad of the wave issues its two block loads and the
 *     prefetch of its next interval-list entry together: one memory round trip per step;
 *   - the read sits in LDS as 4-bit codes; the interval being extended, the first entry of each list and
 *     the prefetched next entry sit in registers.
 * Result identity with the nested form: SMEMs are appended to the read's output when the backward pass
 * emits them (the caller's length filter applied there; the "starts left of the previous one" test needs
 * only the previous start); the list is sorted by (start,end) afterwards (ssg_k_smem_sort), and entries
 * with equal (start,end) describe the same substring, i.e. are identical records, so the sorted list does
 * not depend on insertion order.  The forward list is walked from its top instead of being reversed.
 */
#define SSG_SM_QWORDS 32   /* 8 bases per word: reads up to 256 bases */
enum { SM_FWD = 0, SM_BWD, SM_P3F, SM_READ, SM_P1, SM_P2, SM_P3, SM_OUT, SM_FIN };   /* the three hot states first: the compiler lowers the dispatch to a comparison tree over the value */
enum { SM_PEND_NONE = 0, SM_PEND_FWD, SM_PEND_BWD, SM_PEND_P3 };

/* interval-list entries in scratch and in the carried registers: 16 bytes (x0, x1, x2 < 2^40; info = end position < 256) */
struct alignas(16) ssg_pk_t { uint64_t w0, w1; };
SSG_DEVFN ssg_pk_t ssg_pk(const ssg_intv_t &v)
{ ssg_pk_t p; p.w0 = v.x0 | (v.x1 & 0xffffffull) << 40; p.w1 = (v.x1 >> 24) | v.x2 << 16 | v.info << 56; return p; }
SSG_DEVFN ssg_intv_t ssg_unpk(const ssg_pk_t &p)
{ ssg_intv_t v; v.x0 = p.w0 & 0xffffffffffull; v.x1 = (p.w0 >> 40) | (p.w1 & 0xffffull) << 24; v.x2 = (p.w1 >> 16) & 0xffffffffffull; v.info = p.w1 >> 56; return v; }

#ifndef SSG_SMQ_WAVES
#define SSG_SMQ_WAVES 4
#endif
#ifndef SSG_SMQ_TRIPS
#define SSG_SMQ_TRIPS 2
#endif
/* LPR = lanes per read: 4 (cooperative rank-block fetch) or 1 (each lane fetches whole blocks; 4x fewer wave instructions per read,
 * 4x more translation work per line -- see tools/dbg/gather_probe.cpp for where that starts to matter) */
/* SSG_SMQ_PROBE (diagnostic builds only, tools/dbg/smem_variants.sh): two unused trailing arguments -- the kernarg layout of the round-3
 * builds whose GPU results were wrong although nothing they execute differs */
#ifdef SSG_SMQ_PROBE
#define SSG_SMQ_EXTRA_PARAM , const void *probe_p, int probe_i
#define SSG_SMQ_EXTRA_ARG , (const void*)0, 0
#else
#define SSG_SMQ_EXTRA_PARAM
#define SSG_SMQ_EXTRA_ARG
#endif
template <int LPR>
__global__ void __launch_bounds__(64, SSG_SMQ_WAVES) ssg_k_smem_quad(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const int32_t *read_ids,
                           const uint8_t *seq, const int64_t *off,
                           ssg_intv_t *out_intv, int32_t *out_n, int cap,
                           ssg_intv_t *scratch, int scap, unsigned long long *n_extend, unsigned int *next_read SSG_SMQ_EXTRA_PARAM)
{
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
#define SM_DO_P3() do { while (x < len && SMQ(x) > 3) ++x; if (x >= len) state = SM_OUT; else { ssg_set_intv(ix, SMQ(x), ik); i = x + 1; state = SM_P3F; } } while (0)
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
		if (state == SM_FIN) break;
		if (pend == SM_PEND_NONE) continue;
		/* ---- the one extension site: two rank-block quarters per lane + the next list entry ---- */
		ssg_wave_ldssync();   /* list entries stored by lane 0 of the quad last iteration are read by all four below (same wave: in order on the GPU) */
		const ssg_pk_t *const prev = flip ? vec0 : vec1;
		const int back = pend == SM_PEND_BWD;
		const int jn = back && j + 1 < prev_n ? j + 1 : 0;
		ssg_pk_t pf; pf.w0 = pf.w1 = 0;
		if (jn) pf = SMV(prev, prev_rev ? prev_n - 1 - jn : jn);   /* issued together with the rank-block loads below */
		const ssg_intv_t okc = LPR == 4 ? ssg_bwt_extend1_quad(ix, back ? p : ik, e_c, back, ql) : ssg_bwt_extend1_lean(ix, back ? p : ik, e_c, back);
		++my_nx;
		{
			ssg_pk_t *const curr = flip ? vec1 : vec0;
			if (pend == SM_PEND_FWD) {
				if (okc.x2 != ik.x2) {
					if (curr_n < scap) { if (QW) SMV(curr, curr_n) = ssg_pk(ik); } else ovf = 1;
					++curr_n;
					if (okc.x2 < min_intv) { SM_DO_FWDEND(); pend = SM_PEND_NONE; continue; }   /* break: ik stays the last pushed */
				}
				ik = okc; ik.info = (uint64_t)(i + 1); ++i;
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
/* one lane per read: intervals by (start,end), upstream's ks_introsort(mem_intv) */
__global__ void __launch_bounds__(64) ssg_k_smem_sort(int n_reads, ssg_intv_t *intv, const int32_t *n_intv, int cap)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const int n = n_intv[r];
	if (n > 1 && n <= cap) ssg_introsort(intv + r * cap, (long)n, ssg_intv_lt());
}

/* number of sampled occurrences of one interval (upstream mem_chain: step/count rule) */
SSG_DEVFN int ssg_intv_nocc(const ssg_mem_opt_t &opt, uint64_t x2)
{
	uint64_t step = x2 > (uint64_t)opt.max_occ ? x2 / (uint64_t)opt.max_occ : 1;
	uint64_t cnt = (x2 + step - 1) / step;
	return (int)(cnt < (uint64_t)opt.max_occ ? cnt : (uint64_t)opt.max_occ);
}

/* per read: total #occurrences over its intervals (for the prefix sum that places seeds) */
__global__ void ssg_k_sal_count(ssg_mem_opt_t opt, int n_reads, const ssg_intv_t *intv, const int32_t *n_intv, int cap, int32_t *n_seed)
{
	long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	int n = n_intv[r], tot = 0;
	const ssg_intv_t *p = intv + r * cap;
	for (int i = 0; i < n; ++i) tot += ssg_intv_nocc(opt, p[i].x2);
	n_seed[r] = tot;
}

/*
 * One lane per (read, interval): walks that interval's sampled occurrences.  Seeds are written
 * at seed_off[r] + (occurrences of earlier intervals) + k, i.e. in upstream's visiting order.
 * Invalid seeds (bns_intv2rid < 0) get len = -1 and are skipped by the chaining kernel.
 */
__global__ void ssg_k_sal(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const ssg_intv_t *intv, const int32_t *n_intv, int cap,
                          const int64_t *seed_off, ssg_seed_t *seeds, int32_t *seed_rid)
{	/* one lane per SEED (sampled occurrence): every lane does one independent <=31-step LF walk, so the
	 * random 64-byte fetches of a wave are 64 independent chains and long intervals cost no tail */
	const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= seed_off[n_reads]) return;
	long lo = 0, hi = n_reads;                       /* last r with seed_off[r] <= g */
	while (lo < hi) { long mid = (lo + hi + 1) >> 1; if (seed_off[mid] <= g) lo = mid; else hi = mid - 1; }
	const long r = lo;
	const ssg_intv_t *p = intv + r * cap;
	long k = g - seed_off[r];
	int ii = 0, c;
	while ((c = ssg_intv_nocc(opt, p[ii].x2)) <= k) { k -= c; ++ii; }
	const ssg_intv_t v = p[ii];
	const uint64_t step = v.x2 > (uint64_t)opt.max_occ ? v.x2 / (uint64_t)opt.max_occ : 1;
	ssg_seed_t s;
	s.rbeg = (int64_t)ssg_bwt_sa(ix, v.x0 + (uint64_t)k * step);
	s.qbeg = (int32_t)(v.info >> 32);
	s.len = s.score = (int)((uint32_t)v.info - (uint32_t)(v.info >> 32));
	s.next = -1;
	seeds[g] = s;
	seed_rid[g] = ssg_intv2rid(ix, s.rbeg, s.rbeg + s.len);
}
/* HBM copy of the suffix array sampled every `new_intv` rows instead of the file's sa_intv (upstream's .sa keeps every 32nd row; a seed
 * located through a denser table walks ~new_intv LF steps instead of ~32).  Each sampled row of the file starts a walk: LF(row with
 * SA = v) is the row with SA = v - 1, so the walk fills every row it passes whose index is a multiple of new_intv and ends at the next
 * sampled row -- the walks partition the LF cycle, every row is visited once (seq_len line fetches in all; asking bwt_sa for each new
 * sample instead walks ~sa_intv steps per sample: 7 x the fetches at 32 -> 4).  Walk lengths are geometric, so lanes take new
 * walks from a counter as they finish (one atomic per wave and refill). */
__global__ void ssg_k_sa_densify_walk(ssg_index_view_t ix, int new_intv, uint64_t *sa_new, unsigned long long n_old, unsigned long long *next)
{
	const uint64_t omask = (uint64_t)ix.sa_intv - 1, nmask = (uint64_t)new_intv - 1;
	int nshift = 0; while ((1 << nshift) < new_intv) ++nshift;
	const int lane = wv_lane();
	bool have = false, done = false; uint64_t r = 0, v = 0;
	for (;;) {
		const unsigned long long need = wv_ballot(!have && !done);
		if (need) {
			const int leader = __ffsll(need) - 1;
			unsigned long long base = 0;
			if (lane == leader) base = atomicAdd(next, (unsigned long long)__popcll(need));
			base = (unsigned long long)wv_bcast64((long long)base, leader);
			if (!have && !done) {
				const unsigned long long s = base + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
				if (s >= n_old) done = true;
				else { const uint64_t sv = ix.sa[s]; r = s * (uint64_t)ix.sa_intv; v = s == 0 ? ix.seq_len : sv; sa_new[r >> nshift] = sv; have = true; }
			}
		}
		if (!wv_ballot(have)) break;
		if (have) {
			if (r == ix.primary) r = 0;
			else { const int c = ssg_bwt_sym(ix, r - (r > ix.primary)); r = ix.L2[c] + ssg_occ1(ix, r, c); }
			--v;
			if ((r & omask) == 0) have = false;
			else if ((r & nmask) == 0) sa_new[r >> nshift] = v;
		}
	}
}
#endif
