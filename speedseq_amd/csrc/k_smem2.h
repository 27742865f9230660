/*
 * k_smem2.h -- the seeding kernel of the product path: upstream mem_collect_intv = bwt_smem1a from every
 * position the previous call returned (pass 1), bwt_smem1a with min_intv = occurrences + 1 from the middle of long SMEMs with few
 * occurrences (pass 2), bwt_seed_strategy1 (pass 3); oracle/orc_mem.c restates them from upstream bwt.c / bwamem.c.
 *
 * One lane per read; the three passes are a per-lane state machine around ONE bwt_extend site per loop iteration, so that every lane of
 * the wave that has an extension pending issues its two rank-block fetches (and the fetch of its next interval-list entry) in the same
 * memory round trip, whatever loop of the nested algorithm it is in.  A read's three passes cost ~800 dependent extensions of two random
 * 64-byte lines each; what bounds the kernel is the rate of random lines (tools/dbg/gather_probe.cpp), not streaming bandwidth.
 *
 * Control flow discipline (measured on the MI355X in round 4): the loop has ONE exit and it is wave-uniform
 * (`no lane has work'), and the extension site is one predicated block -- no divergent `break' or `continue' anywhere in the loop.  The
 * round-3 form of this kernel left the loop per lane (`if (state == FIN) break; if (pend == NONE) continue;' plus a `continue' out of the
 * middle of the site); hipcc 7.2 compiled some instances of that text into code that computes different intervals on the GPU (second-pass
 * SMEMs cut short) while the same text is right under the host emulation, and which instance went wrong depended on nothing one would
 * touch on purpose (the translation unit the kernel sat in, two unused arguments).  The uniform form of the same statements is right in
 * every build that was tried.
 *
 * Order of the passes: upstream runs bwt_seed_strategy1 (pass 3) last; here it runs FIRST.  The passes only append to the read's list, pass 2
 * reads pass 1's entries alone, and the list is sorted by (start, end) afterwards, so the order of appending is free -- and pass 3 is a plain
 * chain of ~150 extensions that never makes a read heavy: done first, its seeds survive when the read is given up in pass 1 or 2, and the
 * wave-per-read kernel, whose time is dependent round trips, does not walk that chain again.
 *
 * Per-read budget: a read whose extensions exceed `max_ext', or whose forward pass leaves a row of more than `max_row' intervals, is given up (out_n = -2, nothing of it is kept) and the caller hands it to the
 * wave-per-read kernel: a repeat-heavy read costs 10-50x the average, every one of its extensions is a dependent round trip, and a lane
 * that picked one up late used to hold the whole launch open long after the pool of reads had run dry.
 */
#ifndef SSG_K_SMEM2_H
#define SSG_K_SMEM2_H
#include "ssg_dev.h"

#define SSG_S2_QWORDS 40   /* the read as 4-bit codes, 8 per LDS word: reads up to 320 bases */
enum { S2_FWD = 0, S2_BWD, S2_P3F, S2_READ, S2_P1, S2_P2, S2_P3, S2_OUT, S2_FIN };
enum { S2_PEND_NONE = 0, S2_PEND_FWD, S2_PEND_BWD, S2_PEND_P3 };

/* interval-list entry, 16 bytes: x0, x1 < 2^40, x2 < 2^39 (an occurrence count never exceeds the text length), info = end position < 512 */
struct alignas(16) ssg_pk2_t { uint64_t w0, w1; };
SSG_DEVFN ssg_pk2_t s2_pk(const ssg_intv_t &v)
{ ssg_pk2_t p; p.w0 = v.x0 | (v.x1 & 0xffffffull) << 40; p.w1 = (v.x1 >> 24) | v.x2 << 16 | v.info << 55; return p; }
SSG_DEVFN ssg_intv_t s2_unpk(const ssg_pk2_t &p)
{ ssg_intv_t v; v.x0 = p.w0 & 0xffffffffffull; v.x1 = (p.w0 >> 40) | (p.w1 & 0xffffull) << 24; v.x2 = (p.w1 >> 16) & 0x7fffffffffull; v.info = p.w1 >> 55; return v; }
SSG_DEVFN void s2_set_intv(const ssg_index_view_t &ix, int c, ssg_intv_t &ik)
{	/* upstream bwt_set_intv */
	ik.x0 = ix.L2[c] + 1; ik.x2 = ix.L2[c+1] - ix.L2[c]; ik.x1 = ix.L2[3-c] + 1; ik.info = 0;
}

/* launch statistics of the instrumented instance (TUNE): [0] first lane started, [1] first lane that found the pool of reads empty, [2] last
 * lane done (100 MHz wall clock); [3] reads given up; [8 + b] reads whose extension count has b significant bits; [48] wave rounds,
 * [49] lanes with an extension pending, summed over rounds, [50] lanes with a read */
#ifdef SSG_EMU
static unsigned long long ssg_s2_stat[64];
SSG_DEVFN unsigned long long s2_wall() { return 1; }
#else
__device__ unsigned long long ssg_s2_stat[64];
SSG_DEVFN unsigned long long s2_wall() { return (unsigned long long)wall_clock64(); }
#endif
#define S2_STAT_MIN(i, v) atomicMin(&ssg_s2_stat[i], (unsigned long long)(v))
#define S2_STAT_MAX(i, v) atomicMax(&ssg_s2_stat[i], (unsigned long long)(v))
#define S2_STAT_ADD(i, v) atomicAdd(&ssg_s2_stat[i], (unsigned long long)(v))

#ifndef SSG_S2_WAVES
#define SSG_S2_WAVES 4
#endif
#ifndef SSG_S2_TRIPS
#define SSG_S2_TRIPS 2
#endif

/* ---- uniform values: every lane of the wave holds the same value; tell the compiler (scalar registers, scalar branches) ---- */
#ifdef SSG_EMU
SSG_DEVFN int wv_uni(int v) { return v; }
SSG_DEVFN uint64_t wv_uni64(uint64_t v) { return v; }
#else
SSG_DEVFN int wv_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
SSG_DEVFN uint64_t wv_uni64(uint64_t v) { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32; }
#endif
SSG_DEVFN ssg_intv_t wv_uni_intv(const ssg_intv_t &v) { ssg_intv_t o; o.x0 = wv_uni64(v.x0); o.x1 = wv_uni64(v.x1); o.x2 = wv_uni64(v.x2); o.info = wv_uni64(v.info); return o; }

/*
 * ssg_k_smem_heavy -- one WAVE per read, for the reads the lane kernel gave up (repeat-heavy: thousands of extensions, nearly all of them
 * in the backward passes of bwt_smem1a, where every entry of a long list is extended by the same base).  The wave walks upstream's
 * nested loops as written; what is wave-wide is the inner loop of the backward pass: lane l extends entry base + l of the row, and the
 * row's survivors are appended in list order by ballot + rank.  Upstream appends a survivor when its occurrence count differs from the
 * last appended one's; along a row the counts are non-decreasing (each entry's pattern is a prefix of the one before it), so that is the
 * same as `differs from the previous survivor's'.  Of the dead entries only the row's first can be an SMEM (upstream's test on the start of
 * the call's previous SMEM fails for all the others once it was decided for the first).  The forward extensions and the third pass are
 * chains of dependent extensions and run uniformly on all lanes (same addresses: one fetch per wave).
 * SC = list capacity (> longest read + 1), lists and the read in LDS.  ids: the reads to do (positions in the batch), n_ids on the device.
 */
/* one read by the whole wave (see ssg_k_smem_heavy); qb: >= 264 bytes of LDS for the read, l0 / l1: two lists of > len + 1 entries in LDS.
 * Returns the read's extension count (uniform). */
SSG_DEVFN unsigned long long s2_wave_read(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const int split_len, const int it, const int32_t *read_ids, const uint8_t *seq, const int64_t *off,
                                          ssg_intv_t *out_intv, int32_t *out_n, const int cap, uint8_t *qb, ssg_pk2_t *l0, ssg_pk2_t *l1)
{
	const int lane = wv_lane();
	unsigned long long nx = 0;
	const int r = read_ids ? read_ids[it] : it;
	const uint8_t *q = seq + off[r];
	const int len = (int)(off[r+1] - off[r]);
	ssg_wave_ldssync();
	for (int b = lane; b < len; b += 64) qb[b] = q[b];
	ssg_wave_ldssync();
	ssg_intv_t *const mem = out_intv + (long)it * cap;
	/* what the lane kernel left: out_n = -2: nothing; -3 - k: the third pass is done, its k seeds are the list's first entries (k <= cap: it would have reported an overflow otherwise) */
	const int left = -2 - out_n[it];
	const bool skip_p3 = left >= 1;
	const int n_p3 = skip_p3 ? left - 1 : 0;
	int mem_n = n_p3, ovf = 0;
	/* one call of upstream bwt_smem1a (max_intv = 0): start sx, smallest occurrence count min_intv; returns where the next call starts */
	auto smem1 = [&](const int sx, uint64_t min_intv) -> int {
		if (min_intv < 1) min_intv = 1;
		ssg_intv_t ik; s2_set_intv(ix, (int)qb[sx], ik); ik.info = (uint64_t)(sx + 1);
		int n = 0, i = sx + 1;
		bool stop = false;
		while (!stop && i < len) {   /* forward: the intervals of [sx, i) at every change of the occurrence count, shortest first */
			const int cq = wv_uni((int)qb[i]);
			if (cq < 4) {
				const ssg_intv_t ok = wv_uni_intv(ssg_bwt_extend1_lean(ix, ik, 3 - cq, 0)); ++nx;
				if (ok.x2 != ik.x2) { l0[n++] = s2_pk(ik); stop = ok.x2 < min_intv; }
				if (!stop) { ik = ok; ik.info = (uint64_t)(i + 1); ++i; }
			} else { l0[n++] = s2_pk(ik); stop = true; }
		}
		if (!stop) l0[n++] = s2_pk(ik);
		const int ret = (int)ik.info;
		int cur = 0, prev_n = n, rev = 1, m1_n = 0, m1_last_beg = 0;
		bool more = true;
		for (int ib = sx - 1; more && ib >= -1; --ib) {   /* backward: every interval of the row by the base left of it */
			const int cb = ib < 0 ? -1 : wv_uni((int)qb[ib]) < 4 ? wv_uni((int)qb[ib]) : -1;
			int curr_n = 0; bool have_last = false; uint64_t last_x2 = 0;
			ssg_wave_ldssync();
			for (int base = 0; base < prev_n; base += 64) {
				const int jj = base + lane; const bool valid = jj < prev_n;
				ssg_intv_t p; p.x0 = p.x1 = p.x2 = p.info = 0;
				if (valid) p = s2_unpk((cur ? l1 : l0)[rev ? prev_n - 1 - jj : jj]);
				ssg_intv_t okc; okc.x0 = okc.x1 = okc.x2 = okc.info = 0;
				if (valid && cb >= 0) okc = ssg_bwt_extend1_lean(ix, p, cb, 1);
				if (cb >= 0) nx += (unsigned long long)(prev_n - base < 64 ? prev_n - base : 64);
				const bool alive = valid && cb >= 0 && okc.x2 >= min_intv;
				if (base == 0 && !wv_get((int)alive, 0)) {   /* the row's first entry dies here: an SMEM if it starts left of the call's previous one */
					const int beg = ib + 1;
					if (m1_n == 0 || beg < m1_last_beg) {
						++m1_n; m1_last_beg = beg;
						const int end0 = wv_get((int)(uint32_t)p.info, 0);
						if (end0 - beg >= opt.min_seed_len) {
							if (mem_n < cap) { if (lane == 0) { ssg_intv_t o = p; o.info |= (uint64_t)beg << 32; mem[mem_n] = o; } } else ovf = 1;
							++mem_n;
						}
					}
				}
				const unsigned long long am = wv_ballot(alive);
				if (am) {
					const unsigned long long below = am & ((1ull << lane) - 1ull);
					const int pl = below ? 63 - __clzll(below) : 0;
					const uint64_t px2 = (uint64_t)wv_shfl64_xor((long long)okc.x2, lane ^ pl);   /* okc.x2 of lane pl */
					const bool push = alive && (below ? okc.x2 != px2 : (!have_last || okc.x2 != last_x2));
					const unsigned long long pm = wv_ballot(push);
					if (push) { ssg_intv_t o = okc; o.info = p.info; (cur ? l0 : l1)[curr_n + __popcll(pm & ((1ull << lane) - 1ull))] = s2_pk(o); }
					curr_n += __popcll(pm);
					have_last = true; last_x2 = (uint64_t)wv_get64((long long)okc.x2, 63 - __clzll(am));
				}
			}
			if (curr_n == 0) more = false;
			else { cur ^= 1; rev = 0; prev_n = curr_n; }
		}
		ssg_wave_ldssync();
		return ret;
	};
	if (len >= opt.min_seed_len) {
		int x = 0;
		while (x < len) { if (wv_uni((int)qb[x]) < 4) x = smem1(x, 1); else ++x; }   /* pass 1 */
		const int old_n = mem_n < cap ? mem_n : cap;
		ssg_wave_memsync();
		for (int k = n_p3; k < old_n; ++k) {   /* pass 2: re-seed from the middle of long SMEMs with few occurrences */
			const ssg_intv_t m = wv_uni_intv(mem[k]);
			const int start = (int)(m.info >> 32), end = (int)(uint32_t)m.info;
			if (end - start >= split_len && m.x2 <= (uint64_t)opt.split_width) {
				const int sx = (start + end) >> 1;
				if (wv_uni((int)qb[sx]) < 4) (void)smem1(sx, m.x2 + 1);
			}
		}
		if (opt.max_mem_intv > 0 && !skip_p3) {   /* pass 3: upstream bwt_seed_strategy1 from every position it returns */
			x = 0;
			while (x < len) {
				if (wv_uni((int)qb[x]) > 3) ++x;
				else {
					ssg_intv_t ik; s2_set_intv(ix, (int)qb[x], ik);
					int i = x + 1, nxt = len; bool stop = false;
					while (!stop && i < len) {
						const int cq = wv_uni((int)qb[i]);
						if (cq < 4) {
							const ssg_intv_t ok = wv_uni_intv(ssg_bwt_extend1_lean(ix, ik, 3 - cq, 0)); ++nx;
							if (ok.x2 < (uint64_t)opt.max_mem_intv && i - x >= opt.min_seed_len) {
								if (ok.x2 > 0) {
									if (mem_n < cap) { if (lane == 0) { ssg_intv_t o = ok; o.info = (uint64_t)x << 32 | (uint64_t)(i + 1); mem[mem_n] = o; } } else ovf = 1;
									++mem_n;
								}
								nxt = i + 1; stop = true;
							} else { ik = ok; ++i; }
						} else { nxt = i + 1; stop = true; }
					}
					x = nxt;
				}
			}
		}
	}
	if (lane == 0) out_n[it] = ovf ? -1 : mem_n;
	return nx;
}

/* ---- table of the intervals of all short patterns (ssg_index.ktab) ----
 * The bidirectional interval of a pattern does not depend on the order in which the pattern was extended to, so the intervals of ALL patterns
 * of 1..K bases can be tabulated once per index: 16 bytes each (x0, x1, x2 in 40 bits), level j (4^j entries, little-endian base-4 pattern
 * code: first base least significant) after the levels below it -- 1.4 GB for K = 13 on a human-size index (built in ~20 ms), 22.9 GB for the K = 15 of ssg_seed.cpp.  A bwt_extend
 * whose RESULT pattern has at most K bases then is one 16-byte load at an address that depends only on the read, instead of two rank blocks
 * (2.5 loads each) and the popcounts; roughly four extensions in ten of a 150-base read are that short. */
SSG_DEVFN long s2_ktab_off(int j) { return (long)(((1ull << (2 * j)) - 4ull) / 3ull); }   /* entries of the levels below j */
/* level j from level j - 1: the four one-base left extensions of every pattern by upstream's own bwt_extend (is_back = 1): an entry is bit for
 * bit what the extension it stands in for returns.  Children of pattern code p are 4p .. 4p + 3. */
__global__ void ssg_k_ktab_level(ssg_index_view_t ix, int j, ssg_pk2_t *tab)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= (1L << (2 * (j - 1)))) return;
	ssg_pk2_t *const out = tab + s2_ktab_off(j) + 4 * p;
	ssg_intv_t ok[4];
	if (j == 1) { for (int c = 0; c < 4; ++c) s2_set_intv(ix, c, ok[c]); }
	else ssg_bwt_extend(ix, s2_unpk(tab[s2_ktab_off(j - 1) + p]), ok, 1);
	for (int c = 0; c < 4; ++c) { ok[c].info = 0; out[c] = s2_pk(ok[c]); }
}
/* self-check (SSG_KTAB_VERIFY=1): every `stride'-th pattern of level j once more, the way the seeding kernel reaches it without the table --
 * from its first base by forward extensions with the kernel's own ssg_bwt_extend1_lean -- and compared */
__global__ void ssg_k_ktab_verify(ssg_index_view_t ix, int j, long stride, const ssg_pk2_t *tab, unsigned long long *bad)
{
	const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, code = t * stride;
	if (code >= (1L << (2 * j))) return;
	ssg_intv_t ik;
	s2_set_intv(ix, (int)(code & 3), ik);
	for (int k = 1; k < j; ++k) ik = ssg_bwt_extend1_lean(ix, ik, 3 - (int)((code >> (2 * k)) & 3), 0);
	const ssg_intv_t e = s2_unpk(tab[s2_ktab_off(j) + code]);
	if (ik.x2 != e.x2 || (ik.x2 && (ik.x0 != e.x0 || ik.x1 != e.x1))) atomicAdd(&bad[j], 1ull);
}
/* the 16 codes from position b of a read held as 4-bit codes, 8 per LDS word (word w of the read at qw[w * stride]), first base lowest */
SSG_DEVFN uint64_t s2_window(const uint32_t *qw, int stride, int b)
{
	const int w0 = b >> 3, sh = (b & 7) << 2;
	const int w1 = w0 + 1 < SSG_S2_QWORDS ? w0 + 1 : SSG_S2_QWORDS - 1, w2 = w0 + 2 < SSG_S2_QWORDS ? w0 + 2 : SSG_S2_QWORDS - 1;
	uint64_t v = ((uint64_t)qw[w0 * stride] | (uint64_t)qw[w1 * stride] << 32) >> sh;
	if (sh) v |= (uint64_t)qw[w2 * stride] << (64 - sh);
	return v;
}
/* table code of the n <= 15 unambiguous bases from position b */
SSG_DEVFN uint32_t s2_code(const uint32_t *qw, int stride, int b, int n)
{
	uint64_t v = s2_window(qw, stride, b);
	v &= 0x3333333333333333ull; v = (v | v >> 2) & 0x0f0f0f0f0f0f0f0full; v = (v | v >> 4) & 0x00ff00ff00ff00ffull;
	v = (v | v >> 8) & 0x0000ffff0000ffffull; v = (v | v >> 16) & 0xffffffffull;
	return (uint32_t)v & ((1u << (2 * n)) - 1u);
}

/*
 * seq: concatenated nt4 codes, off[r]..off[r+1] delimit read r.  out_intv: [n_reads x cap]; out_n: per-read interval count (-1: the
 * per-read capacity or a work list overflowed; -2: given up; -3 - k: given up with the k seeds of the finished third pass in place).  scratch: per launched wave 2 lists x scap entries x
 * 64 lanes of 16 bytes, entry e of lane l at [e * 64 + l] (lanes pushing their e-th entries together write one 1-KB span).
 * next_read: shared counter the lanes take their reads from (evens out the per-read cost).  n_ext_read (optional): extensions per read.
 * heavy_ids / n_heavy (optional): the given-up reads, appended in no particular order, for ssg_k_smem_heavy (launched behind this kernel).
 */
template <bool TUNE, bool KT>
__global__ void __launch_bounds__(64, SSG_S2_WAVES) ssg_k_smem2(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const int32_t *read_ids,
                           const uint8_t *seq, const int64_t *off, ssg_intv_t *out_intv, int32_t *out_n, int cap,
                           ssg_pk2_t *scratch, int scap, unsigned long long *n_extend, unsigned int *next_read, unsigned int max_ext, int max_row, uint32_t *n_ext_read, int32_t *heavy_ids, unsigned int *n_heavy,
                           const ssg_pk2_t *kt_tab, int kt_k /* KT: the table of short-pattern intervals, kt_k < min_seed_len */)
{
	__shared__ uint32_t qlds[SSG_S2_QWORDS * 64];
	const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = (int)(threadIdx.x & 63);
	ssg_pk2_t *const vec0 = scratch + (gt >> 6) * 2 * scap * 64 + lane, *const vec1 = vec0 + (long)scap * 64;
	const uint32_t *const ql_ = qlds + lane;
#define S2Q(i) ((int)((ql_[((i) >> 3) * 64] >> (((i) & 7) << 2)) & 15u))
#define S2V(v, e) ((v)[(long)(e) * 64])
	const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
	unsigned long long my_nx = 0;
	unsigned int rd_nx = 0, nx_p3 = 0;   /* extensions of the current read; of its third pass */
	long it = 0;
	int state = S2_READ, pend = S2_PEND_NONE;
	int len = 0, x = 0, k = 0, old_n = 0, caller = 0, mem_n = 0, ovf = 0, heavy = 0, n_p3 = -1;   /* n_p3 < 0: the third pass is not finished */
	ssg_intv_t *mem = 0;
	int sx = 0, i = 0, j = 0, curr_n = 0, prev_n = 0, prev_rev = 0, flip = 0, m1_n = 0, m1_last_beg = 0, ret = 0, e_c = 0;
	uint64_t min_intv = 1, last_x2 = 0;
	ssg_intv_t ik, p;
	ik.x0 = ik.x1 = ik.x2 = ik.info = 0; p = ik;
	ssg_pk2_t pn, c0, first; pn.w0 = pn.w1 = 0; c0 = first = pn;
	if (TUNE) S2_STAT_MIN(0, s2_wall());
	/* transitions done where they arise (no state of their own):
	 * the forward list becomes `prev', walked from its top (= ik), ret = end of the longest match; return of bwt_smem1a to its caller */
#define S2_FWDEND() do { ret = (int)ik.info; flip ^= 1; prev_n = curr_n < scap ? curr_n : scap; prev_rev = 1; curr_n = 0; i = sx - 1; j = 0; first = s2_pk(ik); state = S2_BWD; if (prev_n > max_row) { heavy = 1; state = S2_OUT; } } while (0)
	/* next start of the third pass (upstream bwt_seed_strategy1 from every position): skip ambiguous bases, open the interval */
#define S2_P3END() do { n_p3 = mem_n; nx_p3 = rd_nx; x = 0; state = S2_P1; } while (0)   /* the third pass runs FIRST here (see the kernel's comment): on to pass 1 */
/* With the table the first kt_k - 1 extensions of a start collapse into one look-up: upstream records nothing before the pattern has min_seed_len
 * (> kt_k) bases, so only an ambiguous base or the read's end inside the window matters -- the start then moves past it exactly as upstream's
 * loop returns, one window per trip (state S2_P3 comes back here); the skipped bwt_extend calls still count as the algorithm's. */
#define S2_P3START() do { \
		while (x < len && S2Q(x) > 3) ++x; \
		if (x >= len) S2_P3END(); \
		else if (!KT || kt_k < 2) { s2_set_intv(ix, S2Q(x), ik); i = x + 1; state = S2_P3F; } \
		else { \
			const unsigned long long nm_ = s2_window(ql_, 64, x) & 0x4444444444444444ull; \
			int run_ = nm_ ? (int)((__ffsll(nm_) - 1) >> 2) : 16; \
			run_ = run_ > len - x ? len - x : run_; \
			if (run_ >= kt_k) { ik = s2_unpk(kt_tab[s2_ktab_off(kt_k) + (long)s2_code(ql_, 64, x, kt_k)]); i = x + kt_k; my_nx += (unsigned long long)(kt_k - 1); rd_nx += (unsigned int)(kt_k - 1); state = S2_P3F; } \
			else { my_nx += (unsigned long long)(run_ - 1); rd_nx += (unsigned int)(run_ - 1); if (x + run_ >= len) S2_P3END(); else { x += run_ + 1; state = S2_P3; } } \
		} \
	} while (0)
#define S2_RET() do { if (caller == 1) { x = ret; state = S2_P1; } else { ++k; state = S2_P2; } } while (0)
	/* an SMEM leaves the backward pass: upstream keeps it when it starts left of the previous one of the call; the caller's length filter applied here */
#define S2_EMIT(pp, beg) do { if (m1_n == 0 || (beg) < m1_last_beg) { ++m1_n; m1_last_beg = (beg); \
		if ((int)(uint32_t)(pp).info - (beg) >= opt.min_seed_len) { ssg_intv_t o_ = (pp); o_.info |= (uint64_t)(beg) << 32; if (mem_n < cap) mem[mem_n] = o_; else ovf = 1; ++mem_n; } } } while (0)
	for (;;) {
		/* a bounded number of state-machine steps per extension round: a lane in the middle of a transition sits the round out instead of
		 * making the whole wave walk through the dispatch again */
		/* (a trip after the first is skipped by the whole wave when no lane is in a transition: 52.4 -> 51.4 ms) */
		SSG_UNROLL for (int trip = 0; trip < SSG_S2_TRIPS; ++trip) if ((trip == 0 || wv_ballot(pend == S2_PEND_NONE && state != S2_FIN)) && pend == S2_PEND_NONE && state != S2_FIN) {
			ssg_pk2_t *const curr = flip ? vec1 : vec0;
			if (state == S2_FWD) { /* top of upstream's forward loop: for (i = x + 1; i < len; ++i) */
				if (i < len && S2Q(i) < 4) { pend = S2_PEND_FWD; e_c = 3 - S2Q(i); }
				else { if (curr_n < scap) S2V(curr, curr_n) = s2_pk(ik); else ovf = 1; ++curr_n; S2_FWDEND(); }
			} else if (state == S2_BWD) { /* for (i = x - 1; i >= -1; --i) for (j = 0; j < prev->n; ++j) */
				if (j >= prev_n) {
					bool back_to_caller = curr_n == 0;
					if (!back_to_caller) { flip ^= 1; prev_n = curr_n < scap ? curr_n : scap; prev_rev = 0; curr_n = 0; j = 0; --i; first = c0; back_to_caller = i < -1; }
					if (back_to_caller) S2_RET();
				} else {
					p = s2_unpk(j == 0 ? first : pn);
					const int cb = i < 0 ? -1 : S2Q(i) < 4 ? S2Q(i) : -1;
					if (cb >= 0) { pend = S2_PEND_BWD; e_c = cb; }
					else { /* no base to extend with: only the first (longest) interval of the row can be an SMEM, the rest are no-ops */
						if (j == 0) S2_EMIT(p, i + 1);
						j = prev_n;
					}
				}
			} else if (state == S2_P3F) {
				if (i >= len) S2_P3END();
				else if (S2Q(i) < 4) { pend = S2_PEND_P3; e_c = 3 - S2Q(i); }
				else { x = i + 1; S2_P3START(); }
			} else if (state == S2_READ) {
				it = (long)atomicAdd(next_read, 1u);
				if (it >= n_reads) { state = S2_FIN; if (TUNE) S2_STAT_MIN(1, s2_wall()); }
				else {
					const int r = read_ids ? read_ids[it] : (int)it;
					const uint8_t *q = seq + off[r];
					len = (int)(off[r+1] - off[r]);
					mem = out_intv + (long)it * cap; mem_n = 0; ovf = 0; rd_nx = 0; nx_p3 = 0; heavy = 0;
					/* the read as 4-bit codes, 8 per LDS word, fetched as aligned 8-byte words, four LDS words per round trip (this runs with few lanes active) */
					const unsigned al = (unsigned)((uintptr_t)q & 7), sh8 = al << 3;
					const uint64_t *const qa = (const uint64_t*)(q - al);
					const int nw = (len + 7) >> 3, nb = (int)al + len;   /* nb: bytes from qa to the read's end */
					for (int w0 = 0; w0 < nw; w0 += 4) {
						uint64_t t[5];
						SSG_UNROLL for (int jj = 0; jj < 5; ++jj) t[jj] = (w0 + jj) * 8 < nb ? qa[w0 + jj] : 0;
						SSG_UNROLL for (int jj = 0; jj < 4; ++jj) {
							const int w = w0 + jj;
							if (w < nw) {
								uint64_t v = sh8 ? (t[jj] >> sh8) | (t[jj + 1] << (64 - sh8)) : t[jj];
								v &= 0x0f0f0f0f0f0f0f0full;
								v = (v | v >> 4) & 0x00ff00ff00ff00ffull;
								v = (v | v >> 8) & 0x0000ffff0000ffffull;
								uint32_t v32 = (uint32_t)(v | v >> 16);
								const int nv = len - w * 8;
								if (nv < 8) v32 &= (1u << (nv << 2)) - 1u;
								qlds[w * 64 + lane] = v32;
							}
						}
					}
					x = 0;
					n_p3 = opt.max_mem_intv > 0 ? -1 : 0;
					state = len < opt.min_seed_len ? S2_OUT : opt.max_mem_intv > 0 ? S2_P3 : S2_P1;
				}
			} else if (state == S2_P1) {
				if (x >= len) { old_n = mem_n < cap ? mem_n : cap; k = n_p3 > 0 ? n_p3 : 0; state = S2_P2; }
				else if (S2Q(x) > 3) ++x;
				else { sx = x; min_intv = 1; caller = 1; state = S2_FWD; m1_n = 0; curr_n = 0; i = sx + 1; s2_set_intv(ix, S2Q(sx), ik); ik.info = (uint64_t)(sx + 1); }
			} else if (state == S2_P2) { /* re-seed from the middle of long SMEMs with few occurrences */
				if (k >= old_n) state = S2_OUT;
				else {
					const ssg_intv_t m = mem[k];
					const int start = (int)(m.info >> 32), end = (int)(uint32_t)m.info;
					if (end - start < split_len || m.x2 > (uint64_t)opt.split_width) ++k;
					else {
						sx = (start + end) >> 1; min_intv = m.x2 + 1; caller = 2; m1_n = 0; curr_n = 0; i = sx + 1;
						if (S2Q(sx) > 3) S2_RET();   /* bwt_smem1a returns at once on an ambiguous base */
						else { s2_set_intv(ix, S2Q(sx), ik); ik.info = (uint64_t)(sx + 1); state = S2_FWD; }
					}
				}
			} else if (state == S2_P3) {
				S2_P3START();
			} else { /* S2_OUT */
				const bool given_up = heavy != 0;
				out_n[it] = ovf ? -1 : given_up ? (n_p3 < 0 ? -2 : -3 - n_p3) : mem_n;   /* given up: the seeds of a finished third pass (the list's first n_p3 entries) stay */
				if (given_up && !ovf && heavy_ids) heavy_ids[atomicAdd(n_heavy, 1u)] = (int32_t)it;
				if (given_up) my_nx -= n_p3 < 0 ? rd_nx : rd_nx - nx_p3;   /* n_extend counts the algorithm's extensions (upstream's own count): the wave kernel counts what it redoes of this read */
				if (n_ext_read) n_ext_read[it] = rd_nx;
				if (TUNE) { S2_STAT_ADD(8 + (64 - __clzll((unsigned long long)(rd_nx | 1u))), 1); if (given_up) S2_STAT_ADD(3, 1); }
				state = S2_READ;
			}
		}
		if (TUNE) { const unsigned long long rdy = wv_ballot(pend != S2_PEND_NONE), alv = wv_ballot(state != S2_FIN); if (lane == 0) { S2_STAT_ADD(48, 1); S2_STAT_ADD(49, __popcll(rdy)); S2_STAT_ADD(50, __popcll(alv)); } }
		if (!wv_ballot(state != S2_FIN)) break;   /* the only exit: no lane of the wave has work */
		/* the pattern an extension ends with: [i, end of p) going left, [start, i] going right; up to kt_k bases its interval is in the table.  The two
		 * ways are separate passes for the wave and each is skipped when no lane wants it (the ballots stand where every lane passes) */
		const int pat_b = pend == S2_PEND_BWD ? i : pend == S2_PEND_FWD ? sx : x, pat_n = (pend == S2_PEND_BWD ? (int)p.info : i + 1) - pat_b;
		const bool by_table = KT && pend != S2_PEND_NONE && pat_n <= kt_k;
		const bool any_tab = KT && wv_ballot(by_table) != 0, any_ext = !KT || wv_ballot(pend != S2_PEND_NONE && !by_table) != 0;
		if (pend != S2_PEND_NONE) {
			/* ---- the one extension site: the rank-block quarters of both queries + the next list entry, one memory round trip ---- */
			const ssg_pk2_t *const prev = flip ? vec0 : vec1;
			ssg_pk2_t *const curr = flip ? vec1 : vec0;
			const bool back = pend == S2_PEND_BWD;
			const int jn = back && j + 1 < prev_n ? j + 1 : 0;
			ssg_pk2_t pf; pf.w0 = pf.w1 = 0;
			if (jn) pf = S2V(prev, prev_rev ? prev_n - 1 - jn : jn);   /* issued together with the rank-block loads below */
			ssg_intv_t okc; okc.x0 = okc.x1 = okc.x2 = okc.info = 0;
			if (any_tab) { if (by_table) okc = s2_unpk(kt_tab[s2_ktab_off(pat_n) + (long)s2_code(ql_, 64, pat_b, pat_n)]); }
			if (any_ext) { if (!by_table) okc = ssg_bwt_extend1_lean(ix, back ? p : ik, e_c, back); }
			++my_nx; ++rd_nx;
			if (pend == S2_PEND_FWD) {
				bool fwd_end = false;
				if (okc.x2 != ik.x2) {
					if (curr_n < scap) S2V(curr, curr_n) = s2_pk(ik); else ovf = 1;
					++curr_n;
					if (okc.x2 < min_intv) { S2_FWDEND(); fwd_end = true; }   /* upstream's break: ik stays the last pushed */
				}
				if (!fwd_end) { ik = okc; ik.info = (uint64_t)(i + 1); ++i; }
			} else if (back) {
				pn = pf;
				if (okc.x2 < min_intv) { if (curr_n == 0) S2_EMIT(p, i + 1); }
				else if (curr_n == 0 || okc.x2 != last_x2) {
					ssg_intv_t o = okc; o.info = p.info;
					const ssg_pk2_t po = s2_pk(o);
					if (curr_n == 0) c0 = po;
					if (curr_n < scap) S2V(curr, curr_n) = po; else ovf = 1;
					++curr_n; last_x2 = okc.x2;
				}
				++j;
			} else { /* S2_PEND_P3 */
				if (okc.x2 < (uint64_t)opt.max_mem_intv && i - x >= opt.min_seed_len) {
					if (okc.x2 > 0) { ssg_intv_t o = okc; o.info = (uint64_t)x << 32 | (uint64_t)(i + 1); if (mem_n < cap) mem[mem_n] = o; else ovf = 1; ++mem_n; }
					x = i + 1; S2_P3START();
				} else { ik = okc; ++i; }
			}
			pend = S2_PEND_NONE;
			if (rd_nx > max_ext) { heavy = 1; state = S2_OUT; }   /* given up: the wave-per-read kernel takes the read from its start */
		}
	}
	if (TUNE) S2_STAT_MAX(2, s2_wall());
#undef S2Q
#undef S2V
#undef S2_FWDEND
#undef S2_P3START
#undef S2_P3END
#undef S2_RET
#undef S2_EMIT
	if (n_extend && my_nx) atomicAdd(n_extend, my_nx);
}

#ifdef SSG_HEAVY_MIN_WAVES   /* variant builds: tools/dbg/seed_variants.sh */
#define SSG_HEAVY_BOUNDS __launch_bounds__(64, SSG_HEAVY_MIN_WAVES)
#else
#define SSG_HEAVY_BOUNDS __launch_bounds__(64)
#endif
template <int SC>
__global__ void SSG_HEAVY_BOUNDS ssg_k_smem_heavy(ssg_index_view_t ix, ssg_mem_opt_t opt, const int32_t *ids, const unsigned int *n_ids, const int32_t *read_ids,
                           const uint8_t *seq, const int64_t *off, ssg_intv_t *out_intv, int32_t *out_n, int cap, unsigned long long *n_extend, unsigned int *next)
{
	__shared__ uint8_t qb[328];
	__shared__ ssg_pk2_t lst[2][SC];
	const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
	const int n_todo = (int)*n_ids;
	unsigned long long nx = 0;
	for (;;) {
		const long t = wv_queue_pop(next);
		if (t >= n_todo) break;
		nx += s2_wave_read(ix, opt, split_len, ids[t], read_ids, seq, off, out_intv, out_n, cap, qb, lst[0], lst[1]);
	}
	if (n_extend && nx && wv_lane() == 0) atomicAdd(n_extend, nx);
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
