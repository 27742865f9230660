/*
 * k_chainw.h -- seed chaining and chain filtering for REPEAT-HEAVY reads, one wavefront per read
 * (same functions as k_chain.h: upstream mem_chain / test_and_merge,
 * mem_chain_weight, mem_chain_flt, with identical results).
 *
 * A read that falls in a repeat family brings hundreds to thousands of seeds, and chaining them is
 * order dependent (greedy insertion in seed-visiting order) and quadratic in the filter.  With one
 * lane per read (k_chain.h) these reads -- ~1.5 % of a human-like batch, ~45 % of its seeds --
 * serialise behind HBM-latency pointer chasing.  Here the whole state of one read lives in LDS and
 * the 64 lanes cooperate on every step:
 *   - the chain set is a position-sorted array; the floor lookup is a two-level 64-ary search
 *     (2 LDS reads + 2 ballots), insertion is a wave-parallel shift.  (pos, sec) order of k_chain.h:
 *     among equal positions the first chain stays first, later ones go right behind it, newest
 *     first; the array is also the final "in-order traversal", so no sort is needed afterwards;
 *   - test_and_merge reads the chain's first/last seed summary from LDS;
 *   - chain weights: one lane per chain; the weight sort replays upstream's introsort (ties are the
 *     norm for repeats, and the unstable tie order selects what is kept) on packed keys in LDS;
 *   - the filter tests chain i against 64 kept chains at a time: ballot of the `break' condition,
 *     side effects applied only to the kept chains up to the first break, exactly as the scalar loop.
 * LDS footprint: 31.1 bytes per seed (27 per chain + 4.1 per seed, chains <= seeds); 256 / 1024 / 2048 / 5120 seeds run 20 / 5 / 2 / 1
 * reads per CU.
 */
#ifndef SSG_K_CHAINW_H
#define SSG_K_CHAINW_H
#include "k_chain.h"

template <int CAPC, int CAPS = CAPC> struct ssg_chw_lds_t {   /* CAPC chains, CAPS seeds: the kernels use CAPS = CAPC, so a read that fits can always fall back to the shifting form */
	int64_t a8[CAPC];   /* insertion: rbeg of the chain's last seed [chain id] | weights [chain id] | filter: kept w<<32 | kept sorted idx<<16 | first shadowed */
	int64_t b8[CAPC];   /* insertion: chain position [chain id] (shifting form: positions, sorted [slot]) | sort/filter: w<<32 | chain id, sorted by w */
	int16_t rid[CAPC];  /* insertion: contig of the chain [chain id] (< 32768 contigs: host-checked) | filter: kept state [sorted idx] */
	uint16_t ls[CAPC];  /* last seed [chain id]                                | filter: query end of kept chain */
	uint16_t n[CAPC], fs[CAPC];            /* [chain id]: #seeds (13 bits; bits 13 / 14 / 15 = bit 8 of fq / lq / ll), first seed */
	SSG_DEVMEM int get_fq(int c) const { return fq[c] | (n[c] >> 13 & 1) << 8; }
	SSG_DEVMEM int get_lq(int c) const { return lq[c] | (n[c] >> 14 & 1) << 8; }
	SSG_DEVMEM int get_ll(int c) const { return ll[c] | (n[c] >> 15 & 1) << 8; }
	SSG_DEVMEM int get_n(int c) const { return n[c] & 0x1fff; }
	SSG_DEVMEM void new_chain(int c, int qbeg, int len) { fq[c] = lq[c] = (uint8_t)qbeg; ll[c] = (uint8_t)len; n[c] = (uint16_t)(1 | (qbeg >> 8 & 1) << 13 | (qbeg >> 8 & 1) << 14 | (len >> 8 & 1) << 15); }
	SSG_DEVMEM void add_seed(int c, int qbeg, int len) { lq[c] = (uint8_t)qbeg; ll[c] = (uint8_t)len; n[c] = (uint16_t)((((n[c] & 0x1fff) + 1) & 0x1fff) | (n[c] & 0x2000) | (qbeg >> 8 & 1) << 14 | (len >> 8 & 1) << 15); }
	uint8_t fq[CAPC], lq[CAPC], ll[CAPC];  /* [chain id]: qbeg of first seed, qbeg/len of last seed: the low 8 bits (the ninth of each, for reads of 256..511 bases, rides in n[]) */
	uint16_t ids[CAPS]; /* shifting form of the insertion: chain id of sorted slot (the ranked form needs none: a chain is named after the rank of its first seed) | filter: query begin of kept chain */
	uint16_t nx[CAPS];  /* [seed]: next seed of the same chain */
	uint64_t bm[CAPS / 64];                  /* ranks that hold a chain */
	uint64_t bms[(CAPS / 64 + 63) / 64];     /* words of bm[] that are not empty */
};

/* largest rank <= r whose bit is set, or -1 (r may be -1); wave-uniform, every lane reads the same words */
template <int CAPC, int CAPS> SSG_DEVFN int chw_prev_set(const ssg_chw_lds_t<CAPC, CAPS> &L, int r)
{
	if (r < 0) return -1;
	const int w = r >> 6, bit = r & 63;
	const uint64_t m = L.bm[w] & (bit == 63 ? ~0ull : (1ull << (bit + 1)) - 1);
	if (m) return (w << 6) + 63 - __clzll(m);
	int sw = w >> 6;
	uint64_t sm = L.bms[sw] & ((1ull << (w & 63)) - 1);
	while (!sm && sw > 0) { --sw; sm = L.bms[sw]; }
	if (!sm) return -1;
	const int w2 = (sw << 6) + 63 - __clzll(sm);
	return (w2 << 6) + 63 - __clzll(L.bm[w2]);
}

struct ssg_whi_gt { SSG_DEVMEM bool operator()(int64_t a, int64_t b) const { return (a >> 32) > (b >> 32); } };

/*
 * upstream's introsort of the (w << 32 | id) words in b8[0 .. n) by w, descending (ks_introsort with its unstable tie order: ties are the norm in a
 * repeat family and the order among them decides what the filter keeps), replayed by the whole wave.  One lane running the textbook loops out of
 * LDS pays an LDS round trip per element visit (~100 cycles; 44 % of a 1300-seed read's time).  What the sequential loops do is
 * fixed by the array as it is when a partition starts (A0, after the pivot went to the end):
 *   - the up-scan stops at the positions in (s, t] whose key is <= the pivot's, the down-scan at those in (s, t) whose key is >= it; neither
 *     scan ever reads a position an earlier swap of the same partition wrote, except that the up-scan cannot pass the last swapped j;
 *   - the k-th swap exchanges the k-th up-stop i_k with the k-th down-stop j_k while i_k < j_k; the pivot's place is min(i_(K+1), j_K).
 * So the wave takes 64 positions from each end at a time, finds the stops with two ballots, pairs them by rank through a small table and writes
 * all swaps of the block at once.  Ranges of 16 or fewer stay unsorted as upstream leaves them for its final insertion sort; that pass and depth
 * exhaustion (upstream switches to combsort) stay on one lane.  Scratch: a8[] (dead between the weights and the filter).
 * tests: against ssg_introsort on one lane, same input (ssg_dbg_chain_sort).
 */
template <int CAPC, int CAPS>
SSG_DEVFN void wv_introsort_whi(ssg_chw_lds_t<CAPC, CAPS> &L, const int n)
{
	static_assert(CAPC >= 256, "a8[] holds the pairing tables and the range stack");
	const int lane = wv_lane();
	int64_t *a = L.b8, *tabI = L.a8, *tabJ = L.a8 + 64, *stk = L.a8 + 128;   /* stack entry: s | t << 16 | d << 32 */
	if (n < 2) return;
	if (n == 2) { ssg_wave_ldssync(); if (lane == 0 && (a[1] >> 32) > (a[0] >> 32)) { const int64_t x = a[0]; a[0] = a[1]; a[1] = x; } ssg_wave_ldssync(); return; }
	int d, s = 0, t = n - 1, top = 0;
	for (d = 2; (1 << d) < n; ++d);
	d <<= 1;
	ssg_wave_ldssync();
	for (;;) {
		if (s < t) {
			if (--d == 0) { if (lane == 0) ssg_combsort(a + s, (long)(t - s + 1), ssg_whi_gt()); ssg_wave_ldssync(); t = s; continue; }
			int k = s + ((t - s) >> 1) + 1;
			const int64_t vk = a[k], vi0 = a[s], vj0 = a[t];
			if ((vk >> 32) > (vi0 >> 32)) { if ((vk >> 32) > (vj0 >> 32)) k = t; }
			else k = (vj0 >> 32) > (vi0 >> 32) ? s : t;
			const int64_t rp = k == t ? vj0 : k == s ? vi0 : vk;
			const int P = (int)(rp >> 32);
			ssg_wave_ldssync();
			if (k != t && lane == 0) { a[k] = vj0; a[t] = rp; }
			ssg_wave_ldssync();
			/* the partition */
			int pi = s + 1, pj = t - 1, bi = 0, bj = 0, lastj = 1 << 30, fin;
			unsigned long long MI = 0, MJ = 0; int64_t wi = 0, wj = 0; bool jdone = false;
			for (;;) {
				if (!MI) { bi = pi; pi += 64; const int q = bi + lane; const bool in = q <= t; wi = in ? a[q] : 0; MI = wv_ballot(in && (int)(wi >> 32) <= P); if (!MI) continue; }   /* ends: t is a stop */
				if (!MJ && !jdone) {
					if (pj < s + 1) jdone = true;
					else { bj = pj; pj -= 64; const int q = bj - lane; const bool in = q >= s + 1; wj = in ? a[q] : 0; MJ = wv_ballot(in && (int)(wj >> 32) >= P); if (!MJ) continue; }
				}
				if (jdone) { const int i1 = bi + (int)__builtin_ctzll(MI); fin = i1 < lastj ? i1 : lastj; break; }
				const int cI = __popcll(MI), cJ = __popcll(MJ), c = cI < cJ ? cI : cJ;
				const int rI = wv_rank_of(MI), rJ = wv_rank_of(MJ);
				const bool hasI = MI >> lane & 1, hasJ = MJ >> lane & 1;
				ssg_wave_ldssync();
				if (hasI && rI < c) tabI[rI] = bi + lane;
				if (hasJ && rJ < c) tabJ[rJ] = bj - lane;
				ssg_wave_ldssync();
				const int ti = lane < c ? (int)tabI[lane] : 0, tj = lane < c ? (int)tabJ[lane] : 0;
				const int v = __popcll(wv_ballot(lane < c && ti < tj));   /* the valid pairs are a prefix: i_k ascends, j_k descends */
				if (hasI && rI < v) a[(int)tabJ[rI]] = wi;
				if (hasJ && rJ < v) a[(int)tabI[rJ]] = wj;
				if (v) lastj = wv_get(tj, v - 1);
				if (v < c) { const int i1 = wv_get(ti, v); fin = i1 < lastj ? i1 : lastj; break; }
				MI = wv_ballot(hasI && rI >= c); MJ = wv_ballot(hasJ && rJ >= c);
			}
			ssg_wave_ldssync();
			const int i = fin;
			if (lane == 0) { const int64_t x = a[i]; a[i] = a[t]; a[t] = x; }
			ssg_wave_ldssync();
			if (i - s > t - i) {
				if (i - s > 16) { if (lane == 0) stk[top] = (int64_t)s | (int64_t)(i - 1) << 16 | (int64_t)d << 32; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { if (lane == 0) stk[top] = (int64_t)(i + 1) | (int64_t)t << 16 | (int64_t)d << 32; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) break;
			--top;
			ssg_wave_ldssync();
			const int64_t e = stk[top];
			s = (int)(e & 0xffff); t = (int)(e >> 16 & 0xffff); d = (int)(e >> 32);
		}
	}
	/* upstream's closing insertion sort over the whole array, on one lane: ranges of 16 or fewer were left as they were, and the first element of a
	 * range is never looked at by its partition (it may lie far from home), so no bound on the moves; on an array this close to sorted it costs
	 * about one LDS round trip per element */
	ssg_wave_ldssync();
	if (lane == 0) ssg_insertsort(a, a + n, ssg_whi_gt());
	ssg_wave_ldssync();
}

/* rank: this read's seeds ranked by (reference position, visiting order), or NULL for the shifting form (needs #seeds <= CAPC).
 * Returns 0, or -1 when the ranked form meets what it does not cover (more chains than CAPC, or a third chain at one position:
 * upstream's order among three equal positions is not rank order) -- the caller then chains the read another way. */
template <int CAPC, int CAPS>
SSG_DEVFN int wv_chain_read(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const long r, const int64_t *read_off, const ssg_intv_t *intv,
                             const int32_t *n_intv, int cap, const int64_t *seed_off, const ssg_seed_t *seeds, const int32_t *seed_rid,
                             ssg_chain_t *chains, int32_t *order, int32_t *chain_seeds, int32_t *n_chain, ssg_chw_lds_t<CAPC, CAPS> &L, const uint16_t *rank, int capc_lim, int wave_sort, int32_t *kbflag)
{
	const int lane = wv_lane();
	const int len_read = (int)(read_off[r+1] - read_off[r]);
	const long s0 = seed_off[r]; const int ns = (int)(seed_off[r+1] - s0);
	ssg_chain_t *ch = chains + s0; int32_t *ord = order + s0, *cs = chain_seeds + s0;
	const ssg_seed_t *sd = seeds + s0; const int32_t *srid = seed_rid + s0;
	const int64_t l_pac = ix.l_pac;
	int nc = 0, i, k;
	int n_dup = 0;   /* chains made at a position that had one (wave-uniform): with more than 9 chains the read is flagged for klib's B-tree (k_chain.h ssg_kbflag) */
	/* frac_rep (upstream mem_chain head); wave-uniform */
	int b = 0, e = 0, l_rep = 0, ni = n_intv[r] > 0 ? n_intv[r] : 0;
	const ssg_intv_t *iv = intv + r * cap;
	for (i = 0; i < ni; ++i) {
		int sb = (int)(iv[i].info >> 32), se = (int)(uint32_t)iv[i].info;
		if (iv[i].x2 <= (uint64_t)opt.max_occ) continue;
		if (sb > e) l_rep += e - b, b = sb, e = se;
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	const float frac_rep = (float)l_rep / len_read;
	ssg_wave_ldssync();
	const unsigned long long ph_t0 = ssg_clock();
	int fail = 0;   /* wave-uniform (scalar): set when the ranked form gives the read up; everything after the insertion is skipped then */
	/* ---- greedy chaining in seed-visiting order ---- */
	if (rank) {
		/* The set of chains is a bitmap over position ranks (the universe of possible chain positions is the read's own seeds, ranked
		 * once for the whole batch by a device radix sort): the floor lookup is a few bit scans, a new chain is one bit -- O(1) per
		 * seed instead of a shift of the sorted array (O(chains / 64)), which is what the reads with thousands of seeds paid. */
		for (i = lane; i < (ns + 63) / 64; i += 64) L.bm[i] = 0;
		for (i = lane; i < ((ns + 63) / 64 + 63) / 64; i += 64) L.bms[i] = 0;
		ssg_wave_ldssync();
	}
	if (rank && (wave_sort & 2)) {
		/* 64 seeds at a time, every lane its own: the one-seed-at-a-time loop below is one wave alone on its SIMD running ~200 dependent instructions
		 * per seed with 63 lanes idle (1 460 cycles per seed, profiles/r05n_chain_ab.json).  Here every lane looks up ITS seed's floor chain and decides
		 * (contained / appended / new chain) against the state as it is when the round starts.  A lane's decision is what the sequential loop would
		 * reach unless an EARLIER seed of the round changes what it read: a new chain whose rank falls between its floor and its own rank, or a seed
		 * appended to its floor chain.  The lanes before the first such lane are right: they commit together (their writes touch different chains),
		 * and the next round starts at that lane.  A seed that meets a chain at its own position (upstream's equal-position order) only commits as
		 * the first of a round.  Rounds per 64 seeds: a few once the read's repeat copies have their chains, many while they are being created. */
		for (int i0 = 0; i0 < ns && !fail; i0 += 64) {
			const int me = i0 + lane;
			int64_t my_rbeg = 0; int my_q = 0, my_len = 0, my_rid = -1, my_rk = 0;
			if (me < ns) { const ssg_seed_t sdd = sd[me]; my_rbeg = sdd.rbeg; my_q = sdd.qbeg; my_len = sdd.len; my_rid = srid[me]; my_rk = rank[me]; }
			const int cn = ns - i0 < 64 ? ns - i0 : 64;
			int done = 0;
			while (done < cn) {
				const bool act = lane >= done && lane < cn && my_rid >= 0;
				int fl = -1, res = 1, two_eq = 0, eqp = 0; unsigned ls = 0, n16 = 0;
				if (act) {
					fl = chw_prev_set(L, my_rk); res = 0;
					if (fl >= 0) { /* upstream test_and_merge against the floor chain */
						int64_t f_rbeg = L.b8[fl], l_rbeg = L.a8[fl];
						int f_q = L.fq[fl], l_q = L.lq[fl], l_len = L.ll[fl], crid = L.rid[fl];
						n16 = L.n[fl]; ls = L.ls[fl];
						if (f_rbeg == my_rbeg) {   /* a chain at this very position: upstream tests the FIRST of them */
							eqp = 1;
							const int f2 = chw_prev_set(L, fl - 1);
							if (f2 >= 0 && L.b8[f2] == my_rbeg) {
								fl = f2; two_eq = 1;
								f_rbeg = L.b8[fl]; l_rbeg = L.a8[fl]; n16 = L.n[fl]; ls = L.ls[fl]; f_q = L.fq[fl]; l_q = L.lq[fl]; l_len = L.ll[fl]; crid = L.rid[fl];
							}
						}
						f_q |= (int)(n16 >> 13 & 1) << 8; l_q |= (int)(n16 >> 14 & 1) << 8; l_len |= (int)(n16 >> 15 & 1) << 8;
						if (my_rid != crid) res = 0;
						else if (my_q >= f_q && my_q + my_len <= l_q + l_len && my_rbeg >= f_rbeg && my_rbeg + my_len <= l_rbeg + l_len) res = 1;
						else if ((l_rbeg < l_pac || f_rbeg < l_pac) && my_rbeg >= l_pac) res = 0;
						else {
							const int64_t x = my_q - l_q, y = my_rbeg - l_rbeg;
							if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - l_len < opt.max_chain_gap && y - l_len < opt.max_chain_gap) res = 2;
						}
					}
				}
				/* the first lane an earlier lane of this round interferes with */
				int inv = act && eqp && lane > done;
				int p = cn;
				{
					const unsigned long long b0 = wv_ballot(inv);
					if (b0) p = (int)__builtin_ctzll(b0);
				}
				for (int t = done; t + 1 < p; ++t) {
					const int r1 = wv_get(res, t);
					if (r1 == 1) continue;
					const int rk1 = wv_get(my_rk, t), fl1 = wv_get(fl, t);
					if (act && lane > t) inv |= r1 == 0 ? (fl < rk1 && rk1 <= my_rk) : (fl == fl1);
					const unsigned long long b1 = wv_ballot(inv);
					if (b1) { const int p1 = (int)__builtin_ctzll(b1); p = p1 < p ? p1 : p; }
				}
				const bool com = act && lane < p;
				const unsigned long long cre = wv_ballot(com && res == 0);
				if (wv_ballot(com && res == 0 && (two_eq || nc + wv_rank_of(cre) >= capc_lim))) { fail = 1; break; }   /* not covered here: the caller redoes the read */
				n_dup += __popcll(wv_ballot(com && res == 0 && eqp));
				ssg_wave_ldssync();
				if (com && res == 2) {
					L.nx[ls] = (uint16_t)me; L.ls[fl] = (uint16_t)me; L.a8[fl] = my_rbeg; L.lq[fl] = (uint8_t)my_q; L.ll[fl] = (uint8_t)my_len;
					L.n[fl] = (uint16_t)((((n16 & 0x1fff) + 1) & 0x1fff) | (n16 & 0x2000) | (unsigned)(my_q >> 8 & 1) << 14 | (unsigned)(my_len >> 8 & 1) << 15);
				}
				if (com && res == 0) {
					const int bw = my_rk >> 6;
					atomicOr((unsigned long long*)&L.bm[bw], 1ull << (my_rk & 63)); atomicOr((unsigned long long*)&L.bms[bw >> 6], 1ull << (bw & 63));
					L.b8[my_rk] = my_rbeg; L.a8[my_rk] = my_rbeg; L.new_chain(my_rk, my_q, my_len);
					L.fs[my_rk] = L.ls[my_rk] = (uint16_t)me; L.rid[my_rk] = (int16_t)my_rid;
				}
				ssg_wave_ldssync();
				nc += __popcll(cre);
				done = p;
			}
		}
	} else if (rank) {
		for (int i0 = 0; i0 < ns && !fail; i0 += 64) {
			const int me = i0 + lane;
			int64_t my_rbeg = 0; int my_q = 0, my_len = 0, my_rid = -1, my_rk = 0;
			if (me < ns) { const ssg_seed_t sdd = sd[me]; my_rbeg = sdd.rbeg; my_q = sdd.qbeg; my_len = sdd.len; my_rid = srid[me]; my_rk = rank[me]; }
			const int cn = ns - i0 < 64 ? ns - i0 : 64;
			int pf_t = -1, pf_w = 0; uint64_t pf_m = 0;
			for (int t = 0; t < cn; ++t) {
				const int prid = wv_get(my_rid, t);
				if (prid < 0) continue;
				const int sid = i0 + t, rk = wv_get(my_rk, t);
				const int64_t rbeg = wv_get64(my_rbeg, t);
				const int qbeg = wv_get(my_q, t), len = wv_get(my_len, t);
				/* A chain's id is the rank of its first seed: the floor lookup lands on the chain's state directly.  Two dependent LDS round trips
				 * per seed: the bitmap word of the seed's rank (also the word a new chain sets its bit in), then the floor chain's state in one batch --
				 * and the first of the two is read one seed ahead (a bit this seed sets in that word is patched into the copy). */
				const int bw = rk >> 6, bit = rk & 63;
				const uint64_t mw = pf_t == t ? pf_m : L.bm[bw];
				if (t + 1 < cn) { pf_w = wv_get(my_rk, t + 1) >> 6; pf_m = L.bm[pf_w]; pf_t = t + 1; }   /* the next seed's word travels with this seed's state reads */
				const uint64_t mlow = mw & (bit == 63 ? ~0ull : (1ull << (bit + 1)) - 1);
				int fl = mlow ? (bw << 6) + 63 - __clzll(mlow) : chw_prev_set(L, (bw << 6) - 1), two_equal = 0, res = 0, eqp2 = 0;
				if (fl >= 0) { /* upstream test_and_merge against the floor chain */
					int64_t f_rbeg = L.b8[fl], l_rbeg = L.a8[fl];
					unsigned n16 = L.n[fl], ls = L.ls[fl]; int f_q = L.fq[fl], l_q = L.lq[fl], l_len = L.ll[fl], crid = L.rid[fl];
					if (f_rbeg == rbeg) {   /* a chain at this very position: upstream tests the FIRST of them */
						eqp2 = 1;
						const int f2 = chw_prev_set(L, fl - 1);
						if (f2 >= 0 && L.b8[f2] == rbeg) {
							fl = f2; two_equal = 1;
							f_rbeg = L.b8[fl]; l_rbeg = L.a8[fl]; n16 = L.n[fl]; ls = L.ls[fl]; f_q = L.fq[fl]; l_q = L.lq[fl]; l_len = L.ll[fl]; crid = L.rid[fl];
						}
					}
					f_q |= (int)(n16 >> 13 & 1) << 8; l_q |= (int)(n16 >> 14 & 1) << 8; l_len |= (int)(n16 >> 15 & 1) << 8;
					if (prid != crid) res = 0;
					else if (qbeg >= f_q && qbeg + len <= l_q + l_len && rbeg >= f_rbeg && rbeg + len <= l_rbeg + l_len) res = 1;
					else if ((l_rbeg < l_pac || f_rbeg < l_pac) && rbeg >= l_pac) res = 0;
					else {
						const int64_t x = qbeg - l_q, y = rbeg - l_rbeg;
						if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - l_len < opt.max_chain_gap && y - l_len < opt.max_chain_gap) res = 2;
					}
					if (res == 2) {
						ssg_wave_ldssync();
						if (lane == 0) {
							L.nx[ls] = (uint16_t)sid; L.ls[fl] = (uint16_t)sid; L.a8[fl] = rbeg; L.lq[fl] = (uint8_t)qbeg; L.ll[fl] = (uint8_t)len;
							L.n[fl] = (uint16_t)((((n16 & 0x1fff) + 1) & 0x1fff) | (n16 & 0x2000) | (unsigned)(qbeg >> 8 & 1) << 14 | (unsigned)(len >> 8 & 1) << 15);
						}
						ssg_wave_ldssync();
					}
				}
				/* not covered here: the caller redoes the read.  (Decided on a scalar, outside the lane-conditional regions, so that the exit out of
				 * both loops is a plain scalar branch.) */
				if (wv_get((res == 0 && (two_equal || nc >= capc_lim)) ? 1 : 0, 0)) { fail = 1; break; }
				if (res == 0) { /* new chain; its place in position order is its seed's rank */
					ssg_wave_ldssync();
					if (lane == 0) {
						L.bm[bw] = mw | 1ull << bit;
						if (!mw) L.bms[bw >> 6] |= 1ull << (bw & 63);
						L.b8[rk] = rbeg; L.a8[rk] = rbeg; L.new_chain(rk, qbeg, len);
						L.fs[rk] = L.ls[rk] = (uint16_t)sid; L.rid[rk] = (int16_t)prid;
					}
					ssg_wave_ldssync();
					if (pf_t == t + 1 && pf_w == bw) pf_m |= 1ull << bit;
					++nc; n_dup += wv_get(eqp2, 0);
				}
			}
		}
	} else
	for (int i0 = 0; i0 < ns; i0 += 64) {
		const int me = i0 + lane;
		int64_t my_rbeg = 0; int my_q = 0, my_len = 0, my_rid = -1;
		if (me < ns) { const ssg_seed_t s = sd[me]; my_rbeg = s.rbeg; my_q = s.qbeg; my_len = s.len; my_rid = srid[me]; }
		const int cn = ns - i0 < 64 ? ns - i0 : 64;
		for (int t = 0; t < cn; ++t) {
			const int prid = wv_get(my_rid, t);
			if (prid < 0) continue;
			const int sid = i0 + t;
			const int64_t rbeg = wv_get64(my_rbeg, t);
			const int qbeg = wv_get(my_q, t), len = wv_get(my_len, t);
			/* floor of (rbeg, first): cnt = #chains with pos < rbeg */
			int cnt;
			if (nc <= 64) cnt = __popcll(wv_ballot(lane < nc && L.b8[lane < nc ? lane : 0] < rbeg));
			else {
				const int stride = (nc + 63) >> 6;
				const int e1 = (lane + 1) * stride - 1;
				const int nb = __popcll(wv_ballot(e1 < nc && L.b8[e1 < nc ? e1 : 0] < rbeg));   /* blocks wholly below rbeg */
				const int e2 = nb * stride + lane;
				const int in = lane < stride && e2 < nc;
				cnt = nb * stride + __popcll(wv_ballot(in && L.b8[in ? e2 : 0] < rbeg));
			}
			const int eq = cnt < nc && L.b8[cnt < nc ? cnt : 0] == rbeg;
			const int lower = cnt - 1 + eq;
			int res = 0;
			if (lower >= 0) { /* upstream test_and_merge against the floor chain */
				const int c = L.ids[lower];
				const int64_t f_rbeg = L.b8[lower], l_rbeg = L.a8[c];
				const int f_q = L.get_fq(c), l_q = L.get_lq(c), l_len = L.get_ll(c);
				if (prid != L.rid[c]) res = 0;
				else if (qbeg >= f_q && qbeg + len <= l_q + l_len && rbeg >= f_rbeg && rbeg + len <= l_rbeg + l_len) res = 1;
				else if ((l_rbeg < l_pac || f_rbeg < l_pac) && rbeg >= l_pac) res = 0;
				else {
					const int64_t x = qbeg - l_q, y = rbeg - l_rbeg;
					if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - l_len < opt.max_chain_gap && y - l_len < opt.max_chain_gap) res = 2;
				}
				if (res == 2) {
					ssg_wave_ldssync();
					if (lane == 0) { L.nx[L.ls[c]] = (uint16_t)sid; L.ls[c] = (uint16_t)sid; L.a8[c] = rbeg; L.add_seed(c, qbeg, len); }
					ssg_wave_ldssync();
				}
			}
			if (res == 0) { /* new chain at slot lower+1 */
				const int slot = lower + 1;
				ssg_wave_ldssync();
				for (int hi = nc; hi > slot; hi -= 64) {
					const int lo = hi - 64 > slot ? hi - 64 : slot;
					const int idx = lo + lane;
					int64_t v = 0; uint16_t w = 0;
					if (idx < hi) { v = L.b8[idx]; w = L.ids[idx]; }
					ssg_wave_ldssync();
					if (idx < hi) { L.b8[idx + 1] = v; L.ids[idx + 1] = w; }
				}
				if (lane == 0) {
					L.b8[slot] = rbeg; L.ids[slot] = (uint16_t)nc; L.a8[nc] = rbeg; L.new_chain(nc, qbeg, len);
					L.fs[nc] = L.ls[nc] = (uint16_t)sid; L.rid[nc] = (int16_t)prid;
				}
				ssg_wave_ldssync();
				++nc; n_dup += wv_get(eq, 0);
			}
		}
	}
	int n_out = 0;
	if (!fail) {
	/* ---- upstream mem_chain_weight: one lane per chain ---- */
	const unsigned long long ph_t1 = ssg_clock();
	ssg_wave_ldssync();
	for (int c = lane; c < (rank ? ns : nc); c += 64) {   /* (ranked form: chain ids are the ranks that hold a chain) */
		if (rank && !((L.bm[c >> 6] >> (c & 63)) & 1)) continue;
		int w1 = 0, w2 = 0, sid = L.fs[c], end1 = 0; int64_t end2 = 0;
		const int n = L.get_n(c);
		for (int j = 0; j < n; ++j, sid = L.nx[sid]) {
			const ssg_seed_t s = sd[sid];
			if (s.qbeg >= end1) w1 += s.len; else if (s.qbeg + s.len > end1) w1 += s.qbeg + s.len - end1;
			end1 = end1 > s.qbeg + s.len ? end1 : s.qbeg + s.len;
			if (s.rbeg >= end2) w2 += s.len; else if (s.rbeg + s.len > end2) w2 += (int)(s.rbeg + s.len - end2);
			end2 = end2 > s.rbeg + s.len ? end2 : s.rbeg + s.len;
		}
		int w = w2 < w1 ? w2 : w1;
		L.a8[c] = w < 1<<30 ? w : (1<<30) - 1;
	}
	ssg_wave_ldssync();
	const unsigned long long ph_t2 = ssg_clock();
	/* chains in position order with w >= min_chain_weight -> b8[] as (w, id) */
	int n_chn = 0;
	if (rank) {   /* position order = rank order of the set bits; the (w, id) list goes to b8[], whose positions are no longer needed */
		for (int e0 = 0; e0 < ns; e0 += 64) {
			const int rk = e0 + lane;
			const int set = rk < ns && ((L.bm[rk >> 6] >> (rk & 63)) & 1);
			const int id = rk;
			const int w = set ? (int)L.a8[id] : 0;
			const int keep = set && w >= opt.min_chain_weight;
			const unsigned long long bal = wv_ballot(keep);
			ssg_wave_ldssync();
			if (keep) L.b8[n_chn + wv_rank_of(bal)] = (int64_t)w << 32 | id;
			n_chn += __popcll(bal);
		}
	} else
	for (int e0 = 0; e0 < nc; e0 += 64) {
		const int sl = e0 + lane;
		const int id = sl < nc ? L.ids[sl] : 0;
		const int w = sl < nc ? (int)L.a8[id] : 0;
		const int keep = sl < nc && w >= opt.min_chain_weight;
		const unsigned long long bal = wv_ballot(keep);
		ssg_wave_ldssync();
		if (keep) L.b8[n_chn + wv_rank_of(bal)] = (int64_t)w << 32 | id;
		n_chn += __popcll(bal);
	}
	ssg_wave_ldssync();
	if (n_chn > 0) {
		/* ---- upstream mem_chain_flt ---- */
		const unsigned long long ph_t3 = ssg_clock();
		if ((wave_sort & 1) && n_chn > 24) wv_introsort_whi(L, n_chn); else if (lane == 0) ssg_introsort(L.b8, (long)n_chn, ssg_whi_gt());
		ssg_wave_ldssync();
		const unsigned long long ph_t4 = ssg_clock();
		for (i = lane; i < n_chn; i += 64) L.rid[i] = 0;
		int nk = 0;
		if (wave_sort & 4) {
			/* 64 chains of the sorted list at a time, a lane each, against the kept chains in order.  (One chain at a time against 64 kept chains, below, spends most
			 * of its ~2 200 cycles per chain on the round trips around the few tests it makes.)  A kept chain is one word: query begin | end << 9 | w << 18 in the
			 * high half of a8[k], the first chain it shadows in the low 16 bits.  The sequential loop's order is kept: a lane stops caring at its first `break'
			 * (the tests before it have their side effect, the ones after it none); `first' of a kept chain goes to the smallest i that reaches it with an overlap;
			 * within the block the lanes become kept chains in order, each tested by the later lanes still running. */
			/* The heaviest chains first, all at once: w descends, so a chain that cannot `break' against chain 0 by the weights alone (w < w0 x drop_ratio and w0 - w >=
			 * 2 min_seed_len are both monotone in w) breaks against no chain: these form a prefix [0, m) of the list and are all kept.  What the sequential loop leaves
			 * behind for them: `first' of kept chain e = the first later chain that overlaps it, and large_ovlp of chain e = some earlier chain overlaps it -- two scans
			 * per lane that stop at their first hit (in a repeat family everything overlaps everything: the quadratic loop becomes linear). */
			int m = n_chn;
			{
				const int w0 = (int)(L.b8[0] >> 32);
				for (int i0 = 0; i0 < n_chn; i0 += 64) {
					const int ci = i0 + lane;
					const int wc = ci < n_chn ? (int)(L.b8[ci] >> 32) : 0;
					const unsigned long long um = wv_ballot(ci < n_chn && ((wc < w0 * opt.drop_ratio) & (w0 - wc >= opt.min_seed_len << 1)));
					if (um) { m = i0 + (int)__builtin_ctzll(um); break; }
				}
			}
			ssg_wave_ldssync();
			for (int e = lane; e < m; e += 64) {
				const int64_t me = L.b8[e];
				const int id = (int)(uint32_t)me, wi = (int)(me >> 32);
				L.a8[e] = (int64_t)(((uint64_t)(uint32_t)(L.get_fq(id) | (L.get_lq(id) + L.get_ll(id)) << 9 | wi << 18)) << 32 | 0xffffu);
			}
			ssg_wave_ldssync();
			for (int e0 = 0; e0 < m; e0 += 64) {
				const int e = e0 + lane; const bool act = e < m;
				const int hwe = act ? (int)(L.a8[e] >> 32) : 0;
				const int ib = hwe & 511, ie = hwe >> 9 & 511;
				int f = 0xffff, lo = 0;
				for (int dir = 0; dir < 2; ++dir) {   /* forward: the first later chain of the prefix that overlaps; backward: any earlier one */
					int q = dir ? e - 1 : e + 1;
					bool run = act && (dir ? q >= 0 : q < m);
					while (wv_ballot(run)) {
						if (run) {
							const int hw = (int)(L.a8[q] >> 32);
							const int jb = hw & 511, je = hw >> 9 & 511;
							const int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
							bool ov = false;
							if (e_min > b_max) { const int li = ie - ib, lj = je - jb, min_l = li < lj ? li : lj; ov = e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap; }
							if (ov) { if (dir) lo = 1; else f = q; run = false; }
							else { q += dir ? -1 : 1; run = dir ? q >= 0 : q < m; }
						}
					}
				}
				ssg_wave_ldssync();
				if (act) { L.a8[e] = (int64_t)(((uint64_t)(uint32_t)hwe) << 32 | (uint32_t)f); L.rid[e] = lo ? 2 : 3; }
			}
			nk = m;
			ssg_wave_ldssync();
			for (int i0 = m; i0 < n_chn; i0 += 64) {
				const int ci = i0 + lane; const bool act = ci < n_chn;
				const int64_t me = act ? L.b8[ci] : 0;
				const int id = (int)(uint32_t)me, wi = (int)(me >> 32);
				const int ib = act ? L.get_fq(id) : 0, ie = act ? L.get_lq(id) + L.get_ll(id) : 0;
				int lo = 0, broke = !act, myfirst = 0xffff;
				const int nk0 = nk;
				for (int k0 = 0; k0 < nk0; k0 += 64) {
					if (!wv_ballot(!broke)) break;
					const int kk = k0 + lane;
					const int64_t kw = kk < nk0 ? L.a8[kk] : 0;
					const int kwh = (int)(kw >> 32), kwl = (int)(uint32_t)kw;
					const int cnt = nk0 - k0 < 64 ? nk0 - k0 : 64;
					for (int j = 0; j < cnt; ++j) {
						const int hw = wv_get(kwh, j);
						const int jb = hw & 511, je = hw >> 9 & 511, wj = (int)((unsigned)hw >> 18);
						int ov = 0, brk = 0;
						const int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
						if (!broke && e_min > b_max) {
							const int li = ie - ib, lj = je - jb, min_l = li < lj ? li : lj;
							if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) {
								ov = 1;
								brk = (wi < wj * opt.drop_ratio) & (wj - wi >= opt.min_seed_len << 1);
							}
						}
						const unsigned long long ovm = wv_ballot(ov);
						if (ovm) {
							lo |= ov; broke |= brk;
							if ((wv_get(kwl, j) & 0xffff) == 0xffff && lane == 0) L.a8[k0 + j] = (int64_t)(((uint64_t)(uint32_t)hw << 32) | (uint32_t)(i0 + (int)__builtin_ctzll(ovm)));
						}
					}
				}
				const int cnt_b = n_chn - i0 < 64 ? n_chn - i0 : 64;
				for (int t = 0; t < cnt_b; ++t) {
					if (wv_get(broke, t)) continue;
					const int jb = wv_get(ib, t), je = wv_get(ie, t), wj = wv_get(wi, t);
					int ov = 0, brk = 0;
					const int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
					if (!broke && lane > t && e_min > b_max) {
						const int li = ie - ib, lj = je - jb, min_l = li < lj ? li : lj;
						if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) {
							ov = 1;
							brk = (wi < wj * opt.drop_ratio) & (wj - wi >= opt.min_seed_len << 1);
						}
					}
					const unsigned long long ovm = wv_ballot(ov);
					if (ovm) { lo |= ov; broke |= brk; if (lane == t) myfirst = i0 + (int)__builtin_ctzll(ovm); }
				}
				const unsigned long long keptm = wv_ballot(!broke);
				ssg_wave_ldssync();
				if (!broke) {
					L.a8[nk + wv_rank_of(keptm)] = (int64_t)(((uint64_t)(uint32_t)(ib | ie << 9 | wi << 18)) << 32 | (uint32_t)myfirst);
					L.rid[ci] = lo ? 2 : 3;
				}
				nk += __popcll(keptm);
				ssg_wave_ldssync();
			}
		} else
		for (i = 0; i < n_chn; ++i) {
			const int64_t me = L.b8[i];
			const int id = (int)(uint32_t)me, wi = (int)(me >> 32);
			const int ib = L.get_fq(id), ie = L.get_lq(id) + L.get_ll(id);
			int large_ovlp = 0, broke = 0;
			for (int k0 = 0; k0 < nk && !broke; k0 += 64) {
				const int kk = k0 + lane;
				int ov = 0, brk = 0;
				if (kk < nk) {
					const int jb = L.ids[kk], je = L.ls[kk], wj = (int)(L.a8[kk] >> 32);
					const int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
					if (e_min > b_max) {
						const int li = ie - ib, lj = je - jb, min_l = li < lj ? li : lj;
						if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) {
							ov = 1;
							brk = (wi < wj * opt.drop_ratio) & (wj - wi >= opt.min_seed_len << 1);
						}
					}
				}
				const unsigned long long bb = wv_ballot(brk);
				const int fb = bb ? __builtin_ctzll(bb) : 64;   /* the scalar loop stops at the first break */
				const int act = ov && lane <= fb;
				if (act && (L.a8[kk] & 0xffff) == 0xffff) L.a8[kk] = (L.a8[kk] & ~(int64_t)0xffff) | i;
				large_ovlp |= wv_ballot(act) != 0;
				broke = bb != 0;
			}
			if (!broke) {
				ssg_wave_ldssync();
				if (lane == 0) { L.a8[nk] = (int64_t)wi << 32 | (int64_t)i << 16 | 0xffff; L.ids[nk] = (uint16_t)ib; L.ls[nk] = (uint16_t)ie; L.rid[i] = large_ovlp ? 2 : 3; }
				++nk;
			}
			ssg_wave_ldssync();
		}
		for (k = lane; k < nk; k += 64) { const int f = (int)(L.a8[k] & 0xffff); if (f != 0xffff) L.rid[f] = 1; }
		ssg_wave_ldssync();
		if (opt.max_chain_extend <= n_chn) {
			if (lane == 0) {
				for (i = k = 0; i < n_chn; ++i) {
					const int kk = L.rid[i];
					if (kk == 0 || kk == 3) continue;
					if (++k >= opt.max_chain_extend) break;
				}
				for (; i < n_chn; ++i) if (L.rid[i] < 3) L.rid[i] = 0;
			}
			ssg_wave_ldssync();
		}
		const unsigned long long ph_t5 = ssg_clock();
		if (SSG_TUNING && lane == 0 && CAPS >= 2048) { atomicAdd(&ssg_dbg_cyc[8], ph_t1 - ph_t0); atomicAdd(&ssg_dbg_cyc[9], ph_t2 - ph_t1); atomicAdd(&ssg_dbg_cyc[10], ph_t4 - ph_t3); atomicAdd(&ssg_dbg_cyc[11], ph_t5 - ph_t4); atomicAdd(&ssg_dbg_cyc[12], 1ull); atomicAdd(&ssg_dbg_cyc[13], (unsigned long long)ns); atomicAdd(&ssg_dbg_cyc[14], (unsigned long long)nc); }
		/* ---- survivors in weight order: records, seed lists ---- */
		int pos = 0;
		for (int e0 = 0; e0 < n_chn; e0 += 64) {
			const int sl = e0 + lane;
			const int kept = sl < n_chn ? L.rid[sl] : 0;
			const int64_t me = sl < n_chn ? L.b8[sl] : 0;
			const int id = (int)(uint32_t)me;
			const int n = kept ? L.get_n(id) : 0;
			const unsigned long long bal = wv_ballot(kept != 0);
			const int incl = wv_scan_add(n);
			if (kept) {
				const int start = pos + incl - n;
				int sid = L.fs[id];
				ssg_chain_t c;
				c.pos = sd[sid].rbeg; c.first_seed = (int)(s0 + start); c.last_seed = -1; c.n = n; c.rid = srid[sid];
				c.w = (int)(me >> 32); c.kept = kept; c.first = -1; c.left = c.right = -1; c.frac_rep = frac_rep; c._pad = 0; c.sec = 0;
				ch[id] = c;
				ord[n_out + wv_rank_of(bal)] = id;
				for (int j = 0; j < n; ++j, sid = L.nx[sid]) cs[start + j] = (int)(s0 + sid);
			}
			pos += wv_last(incl);
			n_out += __popcll(bal);
		}
	}
	if (lane == 0) { n_chain[r] = n_out; ssg_kbflag(kbflag, r, nc, n_dup); }
	}
	ssg_wave_ldssync();
	return -fail;
}

/* one wavefront per workgroup; waves pull reads work_order[r_first .. r_end) from a queue.  hrank / hoff: position ranks of the
 * seeds of the heaviest reads (work_order[0 ..)), read k's at hrank[hoff[k] ..]; NULL selects the shifting form.  A read the ranked
 * form gives up (a third chain at one position; capc_lim in the tests) is redone in the shifting form, which covers everything. */
template <int CAP>
__global__ void __launch_bounds__(64) ssg_k_chain_wave(ssg_index_view_t ix, ssg_mem_opt_t opt, int r_first, int r_end,
                            const int64_t *read_off, const ssg_intv_t *intv, const int32_t *n_intv, int cap,
                            const int64_t *seed_off, const ssg_seed_t *seeds, const int32_t *seed_rid,
                            ssg_chain_t *chains, int32_t *order, int32_t *chain_seeds, int32_t *n_chain,
                            const int32_t *work_order, unsigned int *queue, int32_t *kbflag, const uint16_t *hrank, const int64_t *hoff, int capc_lim /* CAP; smaller only in tests */, int wave_sort /* bit 0: the weight sort by the whole wave, bit 1: the insertion 64 seeds a round, bit 2: the filter 64 chains a round (0: one lane / one seed / one chain, A/B and tests) */)
{
	__shared__ ssg_chw_lds_t<CAP, CAP> L;
	for (;;) {
		const long k = r_first + wv_queue_pop(queue);
		if (k >= r_end) break;
		const long r = work_order ? work_order[k] : k;
		int rc = wv_chain_read<CAP, CAP>(ix, opt, r, read_off, intv, n_intv, cap, seed_off, seeds, seed_rid, chains, order, chain_seeds, n_chain, L, hrank ? hrank + hoff[k] : (const uint16_t*)0, capc_lim, wave_sort, kbflag);
		rc = wv_get(rc, 0);
		if (rc) rc = wv_chain_read<CAP, CAP>(ix, opt, r, read_off, intv, n_intv, cap, seed_off, seeds, seed_rid, chains, order, chain_seeds, n_chain, L, (const uint16_t*)0, CAP, wave_sort, kbflag);
	}
}

/* ---- position ranks of the seeds of the heaviest n_heavy reads (work_order[0 .. n_heavy)), for the ranked form ---- */
__global__ void ssg_k_chw_count(int n_heavy, const int32_t *work_order, const int64_t *seed_off, int32_t *hns)
{
	const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (k < n_heavy) { const long r = work_order[k]; hns[k] = (int32_t)(seed_off[r + 1] - seed_off[r]); }
}
/* key = heavy-read index << 34 | reference position (< 2^34); value = slot of the seed in the heavy list; one workgroup per read */
__global__ void ssg_k_chw_keys(int n_heavy, const int32_t *work_order, const int64_t *seed_off, const ssg_seed_t *seeds, const int64_t *hoff, uint64_t *key, uint32_t *val)
{
	const int k = (int)blockIdx.x;
	if (k >= n_heavy) return;
	const long r = work_order[k], s0 = seed_off[r]; const int ns = (int)(seed_off[r + 1] - s0);
	for (int j = (int)threadIdx.x; j < ns; j += (int)blockDim.x) { key[hoff[k] + j] = (uint64_t)k << 34 | (uint64_t)seeds[s0 + j].rbeg; val[hoff[k] + j] = (uint32_t)(hoff[k] + j); }
}
/* after the stable sort by key: sorted place p holds seed val[p] of read key >> 34; its rank is p minus the read's first place */
__global__ void ssg_k_chw_ranks(long n, const uint64_t *key_sorted, const uint32_t *val_sorted, const int64_t *hoff, uint16_t *hrank)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p < n) hrank[val_sorted[p]] = (uint16_t)(p - hoff[key_sorted[p] >> 34]);
}
/* test hook (ssg_dbg_chain_sort): the weight sort of one array by the wave replay (mode 1) or by one lane (mode 0), in the LDS layout of the chaining kernel */
template <int CAP>
__global__ void __launch_bounds__(64) ssg_k_dbg_chain_sort(const int64_t *in, int n, int mode, int64_t *out)
{
	__shared__ ssg_chw_lds_t<CAP, CAP> L;
	const int lane = wv_lane();
	for (int q = lane; q < n; q += 64) L.b8[q] = in[q];
	ssg_wave_ldssync();
	if (mode) wv_introsort_whi(L, n); else if (lane == 0) ssg_introsort(L.b8, (long)n, ssg_whi_gt());
	ssg_wave_ldssync();
	for (int q = lane; q < n; q += 64) out[q] = L.b8[q];
}
#endif
