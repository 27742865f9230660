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
struct ssg_sdp_lds_t { ssg_sdp_key_t key[SSG_SDP_CAP]; uint16_t idx[SSG_SDP_CAP], idx2[SSG_SDP_CAP]; };
#define SSG_SDP_SMALL 96   /* chain extension: most reads end with a handful of regions; bigger sets use the HBM slab */
struct ssg_sdp_small_t { ssg_sdp_key_t key[SSG_SDP_SMALL]; uint16_t idx[SSG_SDP_SMALL], idx2[SSG_SDP_SMALL]; };
struct ssg_sdp_big_t { ssg_sdp_key_t key[SSG_SDP_BIG]; uint16_t idx[SSG_SDP_BIG], idx2[SSG_SDP_BIG]; };
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
SSG_DEVFN int wv_sort_dedup_fast(const ssg_mem_opt_t &opt, int n, ssg_alnreg_t *a, ssg_alnreg_t *tmp, ssg_sdp_key_t *key, uint16_t *idx, uint16_t *idx2, int64_t patch_l_pac = -1)
{
	if (n <= 1) return n;
	const int lane = wv_lane();
	ssg_wave_memsync();
	for (int i = lane; i < n; i += 64) {
		const ssg_alnreg_t r = a[i];
		ssg_sdp_key_t k; k.re = r.re; k.rb = r.rb; k.qb = r.qb; k.qe = r.qe; k.score = r.score; k.rid = r.rid;
		key[i] = k; idx2[i] = (uint16_t)i;
	}
	ssg_wave_memsync();
	if (wv_rank_sort(key, idx2, idx, n, ssg_re_less())) { /* ties in `re`: upstream's unstable sort decides */
		SSG_LANE0(for (int t = 0; t < n; ++t) idx[t] = (uint16_t)t; ssg_key_re_lt lt = { key }; ssg_introsort(idx, (long)n, lt));
	}
	ssg_wave_memsync();
	int n2 = 0;
	if (lane == 0) {
		int i, j, m;
		for (i = 1; i < n; ++i) {
			ssg_sdp_key_t *p = &key[idx[i]];
			if (p->rid != key[idx[i-1]].rid || p->rb >= key[idx[i-1]].re + opt.max_chain_gap) continue;
			for (j = i - 1; j >= 0 && p->rid == key[idx[j]].rid && p->rb < key[idx[j]].re + opt.max_chain_gap; --j) {
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
			if (n2 < 0) break;
		}
		if (n2 == 0) {
			for (i = 0, m = 0; i < n; ++i) if (key[idx[i]].qe > key[idx[i]].qb) idx2[m++] = idx[i];
			n2 = m;
		}
	}
	n2 = wv_bcast(n2, 0);
	if (n2 < 0) return -1;
	ssg_wave_memsync();
	if (wv_rank_sort(key, idx2, idx, n2, ssg_sc_less())) { /* identical (score, rb, qb): tie order selects the survivor */
		SSG_LANE0(for (int t = 0; t < n2; ++t) idx[t] = idx2[t]; ssg_key_sc_lt lt = { key }; ssg_introsort(idx, (long)n2, lt));
	}
	ssg_wave_memsync();
	int m = 0;
	if (lane == 0) {
		int i;
		for (i = 1; i < n2; ++i) {
			const ssg_sdp_key_t x = key[idx[i]], y = key[idx[i-1]];
			if (x.score == y.score && x.rb == y.rb && x.qb == y.qb) key[idx[i]].qe = key[idx[i]].qb;
		}
		for (i = 1, m = 1; i < n2; ++i) if (key[idx[i]].qe > key[idx[i]].qb) idx[m++] = idx[i];
		if (n2 < 1) m = n2;
	}
	m = wv_bcast(m, 0);
	ssg_wave_memsync();
	for (int k = lane; k < m; k += 64) { ssg_alnreg_t r = a[idx[k]]; r.n_comp = 1; tmp[k] = r; }
	ssg_wave_memsync();
	for (int k = lane; k < m; k += 64) a[k] = tmp[k];
	ssg_wave_memsync();
	return m;
}

#endif
