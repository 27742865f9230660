/*
 * k_sdp.h -- upstream mem_sort_dedup_patch on compact keys, shared by the
 * chain-extension kernel (with patch detection) and the mate-rescue kernel (no patching).
 */
#ifndef SSG_K_SDP_H
#define SSG_K_SDP_H
#include "ssg_dev.h"

/* ---------------- region re-sort after a rescue ----------------
 * upstream mem_sort_dedup_patch as mem_matesw calls it (no patching), on 32-byte keys and 16-bit index
 * permutations (LDS for n <= SSG_SDP_CAP, a per-wave HBM slab up to SSG_SDP_BIG) instead of 88-byte
 * records.  Both sorts are done by all 64 lanes as a rank sort (rank = number of smaller keys); when
 * all keys are distinct every correct sort yields upstream's permutation, and when two keys tie the
 * klib introsort is replayed on lane 0 so that the tie order is upstream's.  The redundancy scan and
 * the compactions are short serial passes on lane 0; survivors are gathered by all lanes. */
#define SSG_SDP_CAP 256
#define SSG_SDP_BIG 2048
struct ssg_sdp_key_t { int64_t re, rb; int32_t qb, qe, score, rid; };
struct ssg_sdp_lds_t { ssg_sdp_key_t key[SSG_SDP_CAP]; uint64_t skey[SSG_SDP_CAP]; uint16_t idx[SSG_SDP_CAP], idx2[SSG_SDP_CAP]; };
#define SSG_SDP_SMALL 96   /* chain extension: most reads end with a handful of regions; bigger sets use the HBM slab */
struct ssg_sdp_small_t { ssg_sdp_key_t key[SSG_SDP_SMALL]; uint64_t skey[SSG_SDP_SMALL]; uint16_t idx[SSG_SDP_SMALL], idx2[SSG_SDP_SMALL]; };
struct ssg_sdp_big_t { ssg_sdp_key_t key[SSG_SDP_BIG]; uint64_t skey[SSG_SDP_BIG]; uint16_t idx[SSG_SDP_BIG], idx2[SSG_SDP_BIG]; };
struct ssg_key_re_lt { const ssg_sdp_key_t *k; SSG_DEVMEM bool operator()(uint16_t a, uint16_t b) const { return k[a].re < k[b].re; } };
SSG_DEVFN bool ssg_key_sc_less(const ssg_sdp_key_t &x, const ssg_sdp_key_t &y)
{ return (x.score > y.score) | ((x.score == y.score) & ((x.rb < y.rb) | ((x.rb == y.rb) & (x.qb < y.qb)))); }
struct ssg_key_sc_lt { const ssg_sdp_key_t *k; SSG_DEVMEM bool operator()(uint16_t a, uint16_t b) const { return ssg_key_sc_less(k[a], k[b]); } };

/* rank sort of the n ids in `in` by `less` over key[]; returns (wave-uniform) true when two keys tie,
 * in which case `out` is garbage and the caller replays the exact introsort */
template <class LESS>
SSG_DEVFN bool wv_rank_sort(const ssg_sdp_key_t *key, const uint16_t *in, uint16_t *out, int n, LESS less)
{
	int tie = 0;
	for (int i = wv_lane(); i < n; i += 64) {
		const uint16_t me = in[i]; const ssg_sdp_key_t km = key[me];
		int r = 0, eq = 0;
		for (int j = 0; j < n; ++j) { const ssg_sdp_key_t kj = key[in[j]]; const bool lt = less(kj, km), gt = less(km, kj); r += lt; eq += !(lt | gt); }
		tie |= eq > 1;
		if (eq == 1) out[r] = me;
	}
	return wv_ballot(tie) != 0;
}
/* The same on 64-bit sort keys laid out in input order (skey[t] = key of in[t], every key below 2^64 - 1): a block of 64 keys is loaded once, one per lane, and
 * handed round by v_readlane, so the inner loop touches no memory -- the loop above waits for two dependent loads per key (840 cycles an iteration on the
 * MI355X, measured: three quarters of the re-sort, which is half of ssg_k_chain2aln).  f(t, rank, ties) for every t < n; returns whether any key ties. */
template <class F>
SSG_DEVFN bool wv_rank_u64(const uint64_t *skey, int n, F f)
{
	const int lane = wv_lane();
	int tie = 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int me = i0 + lane;
		const unsigned long long km = me < n ? skey[me] : ~0ull;
		int r = 0, eq = 0;
		for (int j0 = 0; j0 < n; j0 += 64) {
			const unsigned long long kl = j0 + lane < n ? skey[j0 + lane] : ~0ull;
			SSG_UNROLL for (int t = 0; t < 64; ++t) { const unsigned long long k = (unsigned long long)wv_get64((long long)kl, t); r += k < km; eq += k == km; }
		}
		if (me < n) { f(me, r, eq); tie |= eq > 1; }
	}
	return wv_ballot(tie) != 0;
}
struct ssg_re_less { SSG_DEVMEM bool operator()(const ssg_sdp_key_t &a, const ssg_sdp_key_t &b) const { return a.re < b.re; } };
struct ssg_sc_less { SSG_DEVMEM bool operator()(const ssg_sdp_key_t &a, const ssg_sdp_key_t &b) const { return ssg_key_sc_less(a, b); } };

/* upstream mem_patch_reg up to (not including) the global alignment: would this pair of regions be aligned? */
#define SSG_PATCH_MAX_R_BW 0.05f
#define SSG_PATCH_MIN_SC_RATIO 0.90f
SSG_DEVFN int ssg_patch_candidate(const ssg_mem_opt_t &opt, int64_t l_pac, const ssg_sdp_key_t &a, const ssg_sdp_key_t &b)
{
	int w; double r;
	if (a.rb < l_pac && b.rb >= l_pac) return 0;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
	w = (int)((a.re - b.rb) - (a.qe - b.qb));
	w = w > 0 ? w : -w;
	r = (double)(a.re - b.rb) / (b.re - a.rb) - (double)(a.qe - b.qb) / (b.qe - a.qb);
	r = r > 0. ? r : -r;
	if (a.re < b.rb || a.qe < b.qb) { if (w > opt.w << 1 || r >= SSG_PATCH_MAX_R_BW) return 0; }
	else if (w > opt.w << 2 || r >= SSG_PATCH_MAX_R_BW * 2) return 0;
	return 1;
}

/* patch_l_pac < 0: no patching (mem_matesw's call).  Otherwise (mem_align1_core's call) the scan returns -1, with
 * a[] untouched, as soon as a pair of regions would reach mem_patch_reg's global alignment; the caller then runs
 * the general routine.  (Everything mem_patch_reg rejects before aligning leaves no trace, so skipping it is exact.) */
SSG_DEVFN int wv_sort_dedup_fast(const ssg_mem_opt_t &opt, int n, ssg_alnreg_t *a, ssg_alnreg_t *tmp, ssg_sdp_key_t *key, uint64_t *skey, uint16_t *idx, uint16_t *idx2, int64_t patch_l_pac = -1)
{
	if (n <= 1) return n;
	const int lane = wv_lane();
	unsigned long long tq0 = ssg_clock(), tq1;
#define SSG_SDP_PH(x) do { if (SSG_TUNING) { tq1 = ssg_clock(); if (lane == 0) atomicAdd(&ssg_dbg_cyc[64 + (x)], tq1 - tq0); tq0 = tq1; } } while (0)
	ssg_wave_memsync();
	for (int i = lane; i < n; i += 64) {
		const ssg_alnreg_t r = a[i];
		ssg_sdp_key_t k; k.re = r.re; k.rb = r.rb; k.qb = r.qb; k.qe = r.qe; k.score = r.score; k.rid = r.rid;
		key[i] = k; idx2[i] = (uint16_t)i; skey[i] = (uint64_t)r.re;
	}
	ssg_wave_memsync();
	SSG_SDP_PH(0);
	if (wv_rank_u64(skey, n, [&](int t, int rank, int ties) { if (ties == 1) idx[rank] = (uint16_t)t; })) { /* ties in `re`: upstream's unstable sort decides */
		SSG_SDP_PH(1);
		SSG_LANE0(for (int t = 0; t < n; ++t) idx[t] = (uint16_t)t; ssg_key_re_lt lt = { key }; ssg_introsort(idx, (long)n, lt));
		SSG_SDP_PH(2);
		if (SSG_TUNING && lane == 0) { atomicAdd(&ssg_dbg_cyc[72], 1ull); atomicAdd(&ssg_dbg_cyc[73], (unsigned long long)n); }
	} else SSG_SDP_PH(1);
	if (SSG_TUNING && lane == 0) { atomicAdd(&ssg_dbg_cyc[74], 1ull); atomicAdd(&ssg_dbg_cyc[75], (unsigned long long)n); atomicAdd(&ssg_dbg_cyc[76], (unsigned long long)n * n); }
	ssg_wave_memsync();
	/* The redundancy scan is sequential in i (a region excluded by one step is skipped by the later ones), but a step does something only when region i starts within
	 * max_chain_gap of its predecessor's end on the same contig: all lanes look for those i (64 at a time), lane 0 runs upstream's step for them in order.  With the
	 * compactions by ballot and the records copied 8 bytes a lane (88-byte records: a lane per record made eleven strided loads of each): ssg_k_chain2aln 18.0 -> 14.2 ms. */
	int n2 = 0;
	for (int base = 0; base < n && n2 == 0; base += 64) {
		const int i = base + lane;
		bool near = false;
		if (i >= 1 && i < n) { const ssg_sdp_key_t &p = key[idx[i]], &q = key[idx[i-1]]; near = p.rid == q.rid && p.rb < q.re + opt.max_chain_gap; }
		unsigned long long todo = wv_ballot(near);
		if (todo) {
			if (lane == 0) {
				while (todo && n2 == 0) {
					const int ii = base + (int)__builtin_ctzll(todo); todo &= todo - 1;
					ssg_sdp_key_t *p = &key[idx[ii]];
					for (int j = ii - 1; j >= 0 && p->rid == key[idx[j]].rid && p->rb < key[idx[j]].re + opt.max_chain_gap; --j) {
						ssg_sdp_key_t *q = &key[idx[j]];
						int64_t or_, oq, mr, mq;
						if (q->qe == q->qb) continue;
						or_ = q->re - p->rb;
						oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
						mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
						mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
						if (or_ > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq) {
							if (p->score < q->score) { p->qe = p->qb; break; }
							else q->qe = q->qb;
						} else if (patch_l_pac >= 0 && q->rb < p->rb && ssg_patch_candidate(opt, patch_l_pac, *q, *p)) { n2 = -1; break; }
					}
				}
			}
			n2 = wv_bcast(n2, 0);
			ssg_wave_memsync();
		}
	}
	if (n2 == 0) {   /* survivors, in order */
		ssg_wave_memsync();
		for (int base = 0; base < n; base += 64) {
			const int i = base + lane;
			const uint16_t id = i < n ? idx[i] : (uint16_t)0;
			const unsigned long long live = wv_ballot(i < n && key[id].qe > key[id].qb);
			if (i < n && (live >> lane & 1)) idx2[n2 + wv_rank_of(live)] = id;
			n2 += __popcll(live);
		}
	}
	n2 = wv_bcast(n2, 0);
	SSG_SDP_PH(3);
	if (n2 < 0) return -1;
	ssg_wave_memsync();
	/* (score descending, rb, qb) as one 64-bit key when the fields fit (they do: scores below 2^16, positions below 2^38, query offsets below 2^10) */
	int wide = 0;
	for (int t = lane; t < n2; t += 64) {
		const ssg_sdp_key_t &k = key[idx2[t]];
		wide |= k.score < 0 || k.score > 0xfffe || k.rb < 0 || k.rb >= (int64_t)1 << 38 || k.qb < 0 || k.qb > 1023;
		skey[t] = (uint64_t)(0xffff - k.score) << 48 | (uint64_t)k.rb << 10 | (uint64_t)k.qb;
	}
	const bool packed = wv_ballot(wide) == 0;
	ssg_wave_memsync();
	if (packed ? wv_rank_u64(skey, n2, [&](int t, int rank, int ties) { if (ties == 1) idx[rank] = idx2[t]; })
	           : wv_rank_sort(key, idx2, idx, n2, ssg_sc_less())) { /* identical (score, rb, qb): tie order selects the survivor */
		SSG_SDP_PH(4);
		SSG_LANE0(for (int t = 0; t < n2; ++t) idx[t] = idx2[t]; ssg_key_sc_lt lt = { key }; ssg_introsort(idx, (long)n2, lt));
		SSG_SDP_PH(5);
	} else SSG_SDP_PH(4);
	ssg_wave_memsync();
	/* identical hits: the later one of two neighbours goes (the test reads score / rb / qb, the mark is qe: the steps do not depend on one another) */
	for (int i = 1 + lane; i < n2; i += 64) {
		const ssg_sdp_key_t &x = key[idx[i]], &y = key[idx[i-1]];
		if (x.score == y.score && x.rb == y.rb && x.qb == y.qb) key[idx[i]].qe = x.qb;
	}
	ssg_wave_memsync();
	int m = n2 < 1 ? n2 : 1;   /* (the first one stays whatever its state, as upstream's loop from 1 leaves it) */
	for (int base = 1; base < n2; base += 64) {
		const int i = base + lane;
		const uint16_t id = i < n2 ? idx[i] : (uint16_t)0;
		const unsigned long long live = wv_ballot(i < n2 && key[id].qe > key[id].qb);
		ssg_wave_memsync();   /* every lane has read its entry before any is overwritten (targets lie at or below the readers' positions) */
		if (i < n2 && (live >> lane & 1)) idx[m + wv_rank_of(live)] = id;
		m += __popcll(live);
	}
	ssg_wave_memsync();
	{
		static_assert(sizeof(ssg_alnreg_t) % 8 == 0, "ssg_alnreg_t is copied in 8-byte words");
		constexpr int W = (int)(sizeof(ssg_alnreg_t) / 8);
		uint64_t *const tw = (uint64_t*)tmp; uint64_t *const aw = (uint64_t*)a;
		for (int t = lane; t < m * W; t += 64) { const int k = t / W, w = t - k * W; tw[t] = ((const uint64_t*)&a[idx[k]])[w]; }
		ssg_wave_memsync();
		for (int k = lane; k < m; k += 64) tmp[k].n_comp = 1;
		ssg_wave_memsync();
		for (int t = lane; t < m * W; t += 64) aw[t] = tw[t];
		ssg_wave_memsync();
	}
	SSG_SDP_PH(6);
#undef SSG_SDP_PH
	return m;
}

/* 64-bit wave max (every lane gets the result) */
SSG_DEVFN int64_t wv_max64(int64_t v)
{
	for (int d = 1; d < 64; d <<= 1) { const int64_t o = (int64_t)wv_shfl64_xor(v, d); v = v > o ? v : o; }
	return v;
}

/*
 * mem_sort_dedup_patch (no patching) when a[] is the OUTPUT of a previous such call with ONE new region x
 * inserted at a[xpos] -- mem_matesw's situation from its second rescue on.  The previous output is a fixed
 * point of the redundancy scan (every surviving pair inside the chain-gap window was compared and kept),
 * so only pairs with x can change anything, and the scan's verdicts on them follow from values alone:
 *   - x against the regions that end before it (same contig, x.rb < y.re + gap), in descending y.re:
 *     a redundant y with score <= x.score is dropped, the first one with a larger score drops x and stops;
 *   - if x lived, the regions that end after it (y.rb < x.re + gap), in ascending y.re: a redundant y with
 *     score < x.score is dropped, the first one with score >= x.score drops x.
 * Survivors keep their (score, rb, qb) order and x goes to its rank in it.  Whenever the outcome would depend
 * on the order of equal keys (x ties an old region on `re' or on (score, rb, qb), or a dropped candidate
 * ties the stopping region on `re'), -1 is returned with a[] untouched and the caller runs the full sort.
 */
SSG_DEVFN int wv_sort_dedup_incr(const ssg_mem_opt_t &opt, int n, ssg_alnreg_t *a, ssg_alnreg_t *tmp, int xpos)
{
	const int lane = wv_lane();
	ssg_wave_memsync();
	const ssg_alnreg_t x = a[xpos];
	const int64_t gap = opt.max_chain_gap;
	const int64_t NONE_LO = INT64_MIN, NONE_HI = INT64_MAX;
	int tie = 0;
	int64_t ystar = NONE_LO, ycirc = NONE_HI;   /* case 1: largest re that drops x; case 2: smallest re that drops x */
	for (int i = lane; i < n; i += 64) {
		if (i == xpos) continue;
		const ssg_alnreg_t *y = &a[i];
		const int64_t yre = y->re, yrb = y->rb; const int yqb = y->qb, yqe = y->qe, ysc = y->score;
		if (yre == x.re) tie = 1;
		if (ysc == x.score && yrb == x.rb && yqb == x.qb) tie = 1;
		if (y->rid != x.rid) continue;
		if (yre < x.re) { /* p = x, q = y */
			if (x.rb < yre + gap) {
				const int64_t or_ = yre - x.rb, oq = yqb < x.qb ? yqe - x.qb : x.qe - yqb;
				const int64_t mr = yre - yrb < x.re - x.rb ? yre - yrb : x.re - x.rb, mq = yqe - yqb < x.qe - x.qb ? yqe - yqb : x.qe - x.qb;
				if (or_ > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq && x.score < ysc) ystar = ystar > yre ? ystar : yre;
			}
		} else { /* p = y, q = x */
			if (yrb < x.re + gap) {
				const int64_t or_ = x.re - yrb, oq = x.qb < yqb ? x.qe - yqb : yqe - x.qb;
				const int64_t mr = x.re - x.rb < yre - yrb ? x.re - x.rb : yre - yrb, mq = x.qe - x.qb < yqe - yqb ? x.qe - x.qb : yqe - yqb;
				if (or_ > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq && !(ysc < x.score)) ycirc = ycirc < yre ? ycirc : yre;
			}
		}
	}
	if (wv_ballot(tie)) return -1;
	ystar = wv_max64(ystar); ycirc = -wv_max64(-ycirc);
	const bool x_dead1 = ystar != NONE_LO, x_dead = x_dead1 || ycirc != NONE_HI;
	/* second pass: who is dropped, and the new positions */
	int base = 0, bad = 0, x_rank = 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int i = i0 + lane;
		int alive = 0;
		ssg_alnreg_t r;
		if (i < n) {
			r = a[i];
			if (i == xpos) alive = !x_dead;
			else {
				alive = 1;
				if (r.rid == x.rid) {
					if (r.re < x.re) {
						if (x.rb < r.re + gap) {
							const int64_t or_ = r.re - x.rb, oq = r.qb < x.qb ? r.qe - x.qb : x.qe - r.qb;
							const int64_t mr = r.re - r.rb < x.re - x.rb ? r.re - r.rb : x.re - x.rb, mq = r.qe - r.qb < x.qe - x.qb ? r.qe - r.qb : x.qe - x.qb;
							if (or_ > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq && !(x.score < r.score)) {
								if (!x_dead1 || r.re > ystar) alive = 0;
								else if (r.re == ystar) bad = 1;
							}
						}
					} else if (!x_dead1) {
						if (r.rb < x.re + gap) {
							const int64_t or_ = x.re - r.rb, oq = x.qb < r.qb ? x.qe - r.qb : r.qe - x.qb;
							const int64_t mr = x.re - x.rb < r.re - r.rb ? x.re - x.rb : r.re - r.rb, mq = x.qe - x.qb < r.qe - r.qb ? x.qe - x.qb : r.qe - r.qb;
							if (or_ > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq && r.score < x.score) {
								if (ycirc == NONE_HI || r.re < ycirc) alive = 0;
								else if (r.re == ycirc) bad = 1;
							}
						}
					}
				}
			}
		}
		/* survivors keep their (score, rb, qb) order; x (put behind all regions of >= score by mem_matesw) moves to its rank */
		const int old_alive = alive && i != xpos;
		const int before_x = old_alive && ((r.score > x.score) | ((r.score == x.score) & ((r.rb < x.rb) | ((r.rb == x.rb) & (r.qb < x.qb)))));
		const unsigned long long bal = wv_ballot(old_alive);
		if (old_alive) { r.n_comp = 1; tmp[base + wv_rank_of(bal) + (!x_dead && !before_x)] = r; }
		base += __popcll(bal);
		x_rank += __popcll(wv_ballot(before_x));
	}
	if (wv_ballot(bad)) return -1;
	if (!x_dead) { SSG_LANE0(ssg_alnreg_t t = x; t.n_comp = 1; tmp[x_rank] = t); ++base; }
	ssg_wave_memsync();
	for (int k = lane; k < base; k += 64) a[k] = tmp[k];
	ssg_wave_memsync();
	return base;
}

#endif
