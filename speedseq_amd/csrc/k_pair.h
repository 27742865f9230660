/*
 * k_pair.h -- gfx950 kernels for the paired-end stage:
 *
 *   ssg_k_pestat_hist  one lane per pair: upstream mem_pestat's candidate selection (cal_sub,
 *                      mem_infer_dir); insert sizes go into a per-batch, per-orientation histogram
 *                      in HBM (atomicAdd), from which the host finishes quartiles / mean / std in
 *                      double precision in upstream's summation order (sorted order == bin order).
 *   ssg_k_matesw       one wavefront per pair: upstream mem_matesw -- local SW rescue of the mate
 *                      inside the insert-size window (ksw_align2 contract), regions re-sorted and
 *                      de-duplicated after every rescue exactly like upstream.
 *   ssg_k_pair_final   one lane per pair: mem_mark_primary_se, mem_pair, the pairing/MAPQ decision
 *                      tree of mem_sam_pe and mem_reg2sam's record selection; emits "alignment
 *                      requests" (which region becomes which SAM record, with flag and MAPQ) that
 *                      ssg_k_reg2aln turns into CIGAR/NM/MD.
 */
#ifndef SSG_K_PAIR_H
#define SSG_K_PAIR_H
#include <math.h>
#include "k_extend.h"


#include "k_mswlane.h"

SSG_DEVFN int ssg_cal_sub(const ssg_mem_opt_t &opt, const ssg_alnreg_t *a, int n)
{	/* upstream cal_sub */
	int j;
	for (j = 1; j < n; ++j) {
		int b_max = a[j].qb > a[0].qb ? a[j].qb : a[0].qb;
		int e_min = a[j].qe < a[0].qe ? a[j].qe : a[0].qe;
		if (e_min > b_max) {
			int min_l = a[j].qe - a[j].qb < a[0].qe - a[0].qb ? a[j].qe - a[j].qb : a[0].qe - a[0].qb;
			if (e_min - b_max >= min_l * opt.mask_level) break;
		}
	}
	return j < n ? a[j].score : opt.min_seed_len * opt.a;
}

/* hist: [n_batches][4][SSG_MAX_INS_HIST] */
__global__ void ssg_k_pestat_hist(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_pairs, const int64_t *reg_off, const ssg_alnreg_t *regs,
                                  const int32_t *n_reg, const int32_t *pair_batch, uint32_t *hist)
{
	long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pairs) return;
	const ssg_alnreg_t *r0 = regs + reg_off[2*i], *r1 = regs + reg_off[2*i+1];
	int n0 = n_reg[2*i], n1 = n_reg[2*i+1];
	if (n0 == 0 || n1 == 0) return;
	if (ssg_cal_sub(opt, r0, n0) > 0.8 * r0[0].score) return;
	if (ssg_cal_sub(opt, r1, n1) > 0.8 * r1[0].score) return;
	if (r0[0].rid != r1[0].rid) return;
	int64_t is;
	int dir = ssg_infer_dir(ix.l_pac, r0[0].rb, r1[0].rb, &is);
	if (is && is <= opt.max_ins && is < SSG_MAX_INS_HIST)
		atomicAdd(&hist[((long)pair_batch[i] * 4 + dir) * SSG_MAX_INS_HIST + is], 1u);
}

/* ---------------- mate rescue ---------------- */
#define SSG_MS_BCAP 32768   /* rows of a rescue window (b[] entries) per resident wave */

template <bool WIDE>
SSG_DEVFN int wv_matesw(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const ssg_pestat_t *pes, const ssg_alnreg_t a,
                        int l_ms, const uint8_t *ms, ssg_alnreg_t *ma, int *ma_n_, int ma_cap,
                        uint8_t *tbuf, int tcap, uint8_t *revbuf, unsigned long long *bscratch, int *err, unsigned long long *cells, unsigned long long *ph,
                        ssg_alnreg_t *sdp_tmp, ssg_sdp_lds_t *sdp_lds, ssg_sdp_big_t *sdp_big, int *ma_fixed, const ssg_msres_t *jres /* this anchor's four slots, or null */, unsigned long long *npre)
{	/* upstream mem_matesw; jres: forward passes computed ahead of the decision (k_mswlane.h); *ma_fixed: ma[] is the output of an earlier re-sort in this kernel (see wv_sort_dedup_incr); ph[]: cycle counters per phase (fetch, SW, re-sort, window rows) */
	const int64_t l_pac = ix.l_pac;
	int i, r, skip[4], n = 0, rid = -1, ma_n = *ma_n_;
	for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	{	/* a hit already inside the insert-size window of an orientation makes the rescue for it unnecessary: 64 hits per step */
		int seen = 0;
		ssg_wave_memsync();
		for (i = wv_lane(); i < ma_n; i += 64) {
			int64_t dist;
			r = ssg_infer_dir(l_pac, a.rb, ma[i].rb, &dist);
			if (dist >= pes[r].low && dist <= pes[r].high) seen |= 1 << r;
		}
		for (r = 0; r < 4; ++r) if (wv_ballot(seen >> r & 1)) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (r = 0; r < 4; ++r) {
		int is_rev, is_larger, xpos_last = -1;   /* xpos_last: where this window's hit went into ma[] */
		int64_t rb, re;
		if (skip[r]) continue;
		is_rev = (r >> 1 != (r & 1));
		is_larger = !(r >> 1);
		if (!is_rev) {
			rb = is_larger ? a.rb + pes[r].low : a.rb - pes[r].high;
			re = (is_larger ? a.rb + pes[r].high : a.rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a.rb + pes[r].low : a.rb - pes[r].high) - l_ms;
			re = is_larger ? a.rb + pes[r].high : a.rb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) { /* upstream bns_fetch_seq around the window's midpoint */
			int rv; rid = ssg_pos2rid(ix, ssg_depos(ix, (rb + re) >> 1, &rv));
			int64_t far_beg = ix.ctg_off[rid], far_end = far_beg + ix.ctg_len[rid];
			if (rv) { int64_t t2 = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t2; }
			rb = rb > far_beg ? rb : far_beg;
			re = re < far_end ? re : far_end;
		}
		if (a.rid == rid && re - rb >= opt.min_seed_len) {
			if (re - rb > tcap || re - rb > SSG_MS_BCAP) { *err = 1; continue; }
			unsigned long long c0 = ssg_clock();
			const int xtra = SSG_KSW_XSUBO | SSG_KSW_XSTART | (l_ms * opt.a < 250 ? SSG_KSW_XBYTE : 0) | (opt.min_seed_len * opt.a);
			ssg_kswr_t aln;
			int pre = 0;   /* the window's forward pass (1), and its reverse pass too (2), are in its slot (same window: start and length checked) */
			if (jres) pre = wv_get(jres[r].state >= 1 && jres[r].rb == rb && jres[r].tlen == (int)(re - rb) ? jres[r].state : 0, 0);
			if (pre) { aln.score = wv_get(jres[r].score, 0); aln.te = wv_get(jres[r].te, 0); aln.qe = wv_get(jres[r].qe, 0); aln.score2 = wv_get(jres[r].score2, 0); aln.te2 = wv_get(jres[r].te2, 0); aln.tb = aln.qb = -1; ++*npre; if (cells) *cells += (unsigned long long)(re - rb) * l_ms; }
			if (pre == 2) { aln.tb = wv_get(jres[r].tb, 0); aln.qb = wv_get(jres[r].qb, 0); }
			const bool want_rev = !pre || (pre == 1 && ssg_align2_has_rev(xtra, aln.score));   /* only a window that reached minsc has a reverse pass */
			if (want_rev) wv_fetch_ref(ix, rb, pre ? rb + aln.te + 1 : re, tbuf);
			ph[0] += ssg_clock() - c0; c0 = ssg_clock();
			ssg_seqv_t q;
			if (is_rev) {
				if (want_rev) {
					ssg_wave_memsync();
					for (i = wv_lane(); i < l_ms; i += 64) revbuf[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
					ssg_wave_memsync();
				}
				q.p = revbuf; q.dir = 1;
			} else { q.p = ms; q.dir = 1; }
			ssg_seqv_t t = { tbuf, 1 };
			if (!pre) aln = wv_align2<WIDE>(opt, l_ms, q, (int)(re - rb), t, xtra, bscratch, cells);
			else if (want_rev) wv_align2_rev<WIDE>(opt, l_ms, q, t, xtra, aln, bscratch, cells);
			if (SSG_TUNING && wv_lane() == 0) { if (pre == 1 && want_rev) { atomicAdd(&ssg_dbg_cyc[29], 1ull); atomicAdd(&ssg_dbg_cyc[30], (unsigned long long)(aln.te + 1)); } if (!pre) atomicAdd(&ssg_dbg_cyc[31], 1ull); }
			ph[1] += ssg_clock() - c0; if (SSG_TUNING) ph[3] += (unsigned long long)(re - rb);
			if (aln.score >= opt.min_seed_len && aln.qb >= 0) {
				ssg_alnreg_t b;
				b.rb = b.re = 0; b.qb = b.qe = 0; b.truesc = b.sub = b.alt_sc = b.sub_n = b.w = b.secondary_all = b.seedlen0 = b.n_comp = 0; b.frac_rep = 0; b.hash = 0;
				b.rid = a.rid;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score;
				b.csub = aln.score2;
				b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				if (ma_n >= ma_cap) { *err = 2; }
				else { /* insert before the first lower-scoring hit: all lanes search and shift */
					int t2 = ma_n;
					for (int k2 = wv_lane(); k2 < ma_n; k2 += 64) if (ma[k2].score < b.score) { t2 = k2; break; }
					t2 = wv_min(t2);
					if (ma_n - t2 <= SSG_SDP_BIG) {
						ssg_wave_memsync();
						for (int k2 = t2 + wv_lane(); k2 < ma_n; k2 += 64) sdp_tmp[k2 - t2] = ma[k2];
						ssg_wave_memsync();
						for (int k2 = t2 + wv_lane(); k2 < ma_n; k2 += 64) ma[k2 + 1] = sdp_tmp[k2 - t2];
						SSG_LANE0(ma[t2] = b);
					} else {
						SSG_LANE0(for (int k2 = ma_n; k2 > t2; --k2) ma[k2] = ma[k2-1]; ma[t2] = b);
					}
					++ma_n; xpos_last = t2;
				}
			}
			++n;
		}
		if (n) { /* upstream re-sorts after every attempted window once one was tried */
			unsigned long long c1 = ssg_clock(); const int n_in = ma_n;
			int m = -1;
			if (*ma_fixed && ma_n - 1 <= SSG_SDP_BIG) m = xpos_last < 0 ? ma_n : wv_sort_dedup_incr(opt, ma_n, ma, sdp_tmp, xpos_last);
			if (m < 0) m = ma_n <= SSG_SDP_CAP ? wv_sort_dedup_fast(opt, ma_n, ma, sdp_tmp, sdp_lds->key, sdp_lds->skey, sdp_lds->idx, sdp_lds->idx2)
			           : ma_n <= SSG_SDP_BIG ? wv_sort_dedup_fast(opt, ma_n, ma, sdp_tmp, sdp_big->key, sdp_big->skey, sdp_big->idx, sdp_big->idx2)
			           : wv_sort_dedup_patch<WIDE>(ix, opt, 0, 0, ma_n, ma, tbuf, tcap, err, cells);
			ma_n = m; *ma_fixed = 1; xpos_last = -1;
			c1 = ssg_clock() - c1; ph[2] += c1; ph[n_in <= 8 ? 5 : n_in <= 64 ? 6 : 7] += c1;
		}
	}
	*ma_n_ = ma_n;
	return n;
}

/*
 * One wavefront per pair (grid-strided).  regs: per-read slices [reg_off[r], reg_off[r+1]) with
 * head-room for rescued hits; n_reg updated in place.
 */
/* Which pairs can mem_matesw do anything for?  One lane per pair (heaviest-first order): a pair needs the rescue kernel iff
 * some anchor (a hit within pen_unpaired of its read's best, the first max_matesw of them) has an orientation that is neither
 * failed nor already served by a mate hit inside its insert-size window -- the test at the top of upstream mem_matesw, on
 * the lists as they are before any rescue (if no anchor qualifies nothing is ever inserted, so the lists stay as they are).
 * Pairs with long lists are passed on without looking.  The rescue kernel then pops a list of real work instead of 10^6
 * pairs through one atomic counter. */
__global__ void __launch_bounds__(64) ssg_k_matesw_need(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_pairs, const int64_t *reg_off, const ssg_alnreg_t *regs,
                                  const int32_t *n_reg, const int32_t *pair_batch, const ssg_pestat_t *pes_all, const int32_t *work_order,
                                  int32_t *todo_list, unsigned int *n_todo)
{
	const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_pairs) return;
	const long p = work_order ? work_order[g] : g;
	const ssg_pestat_t *pes = pes_all + (long)pair_batch[p] * 4;
	const ssg_alnreg_t *a[2] = { regs + reg_off[2*p], regs + reg_off[2*p+1] };
	const int an[2] = { n_reg[2*p], n_reg[2*p+1] };
	int need = an[0] + an[1] > 32;
	for (int i = 0; i < 2 && !need; ++i) {
		if (an[i] == 0) continue;
		const int thr = a[i][0].score - opt.pen_unpaired;
		int cnt = 0;
		for (int j = 0; j < an[i] && cnt < opt.max_matesw && !need; ++j) {
			if (a[i][j].score < thr) continue;
			++cnt;
			int skip[4];
			for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
			const int64_t arb = a[i][j].rb;
			for (int m = 0; m < an[!i]; ++m) {
				int64_t dist;
				const int r = ssg_infer_dir(ix.l_pac, arb, a[!i][m].rb, &dist);
				if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
			}
			if (skip[0] + skip[1] + skip[2] + skip[3] != 4) need = 1;
		}
	}
	if (need) todo_list[atomicAdd(n_todo, 1u)] = (int32_t)p;
}

template <bool WIDE>
__global__ void __launch_bounds__(256, SSG_SW_WAVES_PER_SIMD) ssg_k_matesw(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_pairs, const uint8_t *seq, const int64_t *read_off,
                             const int64_t *reg_off, ssg_alnreg_t *regs, int32_t *n_reg, const int32_t *pair_batch, const ssg_pestat_t *pes_all,
                             ssg_alnreg_t *bcopy, uint8_t *tglb, unsigned long long *bglb, int32_t *err, unsigned long long *cells, unsigned long long *n_rescue,
                             const int32_t *work_order, unsigned int *queue, ssg_sdp_big_t *sdpbig, const unsigned int *n_todo /* work_order[] holds this many pairs */,
                             const ssg_msres_t *jres, const int64_t *jbase /* forward passes of the windows of pair kq's sides: slots jbase[2 kq + i] + 4 j + r (k_mswlane.h); or null */,
                             const uint8_t *sdp_fixed /* per read: the list is a fixed point of mem_sort_dedup_patch's scan already (k_extend.h); or null */)
{
	__shared__ uint8_t revlds[SSG_WAVES_PER_WG][320];
	__shared__ ssg_sdp_lds_t sdplds[SSG_WAVES_PER_WG];
	const int wslot = (int)(threadIdx.x >> 6);
	const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + wslot;
	uint8_t *tg = tglb + wave0 * (long)SSG_TWIN_GLB;
	unsigned long long *bs = bglb + wave0 * (long)SSG_MS_BCAP;
	ssg_alnreg_t *bc = bcopy + wave0 * (128L + SSG_SDP_BIG);   /* upstream's b[2] (2 x 64) + the re-sort gather buffer */
	unsigned long long nc = 0, nres = 0, npre = 0, ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	const unsigned long long k0 = ssg_clock();
	for (;;) { /* pairs come from a heaviest-first queue (many candidate hits => many rescues) */
		const long kq = wv_queue_pop(queue);
		if (kq >= (n_todo ? (long)*n_todo : (long)n_pairs)) break;
		const long p = work_order ? work_order[kq] : kq;
		const ssg_pestat_t *pes = pes_all + (long)pair_batch[p] * 4;
		int myerr = 0, nb[2] = {0, 0};
		ssg_alnreg_t *a[2] = { regs + reg_off[2*p], regs + reg_off[2*p+1] };
		int an[2] = { n_reg[2*p], n_reg[2*p+1] };
		const int cap[2] = { (int)(reg_off[2*p+1] - reg_off[2*p]), (int)(reg_off[2*p+2] - reg_off[2*p+1]) };
		/* b[i] = hits within pen_unpaired of the best, copied BEFORE any rescue (upstream order) */
		for (int i = 0; i < 2; ++i) {
			int cnt = 0;
			const int thr = an[i] ? a[i][0].score - opt.pen_unpaired : 0;
			for (int j = wv_lane(); j < an[i]; j += 64) if (a[i][j].score >= thr) ++cnt;
			cnt = wv_sum(cnt);
			nb[i] = cnt < opt.max_matesw ? cnt : opt.max_matesw;
			if (nb[i] > 64) { nb[i] = 64; myerr = 3; }
		}
		if (nb[0] + nb[1] > 0) {
			int fixed[2] = { sdp_fixed ? (int)sdp_fixed[2*p] : 0, sdp_fixed ? (int)sdp_fixed[2*p + 1] : 0 };   /* a[i] is the output of a plain re-sort (stage 1's, or this kernel's): the next one is incremental */
			ssg_wave_memsync();
			for (int i2 = 0; i2 < 2; ++i2) { /* the first nb[i2] qualifying hits, in list order: 64 per step */
				const int thr = an[i2] ? a[i2][0].score - opt.pen_unpaired : 0;
				int c2 = 0;
				for (int j0 = 0; j0 < an[i2] && c2 < nb[i2]; j0 += 64) {
					const int j2 = j0 + wv_lane();
					const int f = j2 < an[i2] && a[i2][j2].score >= thr;
					const unsigned long long bal = wv_ballot(f);
					const int at = c2 + wv_rank_of(bal);
					if (f && at < nb[i2]) bc[i2 * 64 + at] = a[i2][j2];
					c2 += __popcll(bal);
				}
			}
			ssg_wave_memsync();
			for (int i = 0; i < 2; ++i)
				for (int j = 0; j < nb[i]; ++j) {
					const int l_ms = (int)(read_off[2*p + !i + 1] - read_off[2*p + !i]);
					const uint8_t *ms = seq + read_off[2*p + !i];
					nres += (unsigned long long)wv_matesw<WIDE>(ix, opt, pes, bc[i * 64 + j], l_ms, ms, a[!i], &an[!i], cap[!i], tg, SSG_TWIN_GLB, revlds[wslot], bs, &myerr, &nc, ph, bc + 128, &sdplds[wslot], sdpbig + wave0, &fixed[!i], jres ? jres + jbase[2*kq + i] + 4 * j : 0, &npre);
				}
		}
		if (wv_lane() == 0) { n_reg[2*p] = an[0]; n_reg[2*p+1] = an[1]; if (myerr) err[p] = myerr; }
	}
	if (wv_lane() == 0) {
		if (cells) atomicAdd(cells, nc);
		if (n_rescue) { atomicAdd(n_rescue, nres); if (npre) atomicAdd(n_rescue + 1, npre); }   /* [1]: windows whose forward pass was there already */
		if (SSG_TUNING) { ph[4] = ssg_clock() - k0; for (int t = 0; t < 8; ++t) atomicAdd(&ssg_dbg_cyc[t], ph[t]); }
	}
}

/* ---------------- primary marking, pairing, MAPQ ---------------- */
SSG_DEVFN uint64_t ssg_hash64(uint64_t key)
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}
struct ssg_reg_hash_lt {
	SSG_DEVMEM bool operator()(const ssg_alnreg_t &a, const ssg_alnreg_t &b) const
	{ return (a.score > b.score) | ((a.score == b.score) & (a.hash < b.hash)); } /* branch-free: see ssg_chain_key_lt */
};

struct ssg_reg_hash_idx_lt {
	const ssg_alnreg_t *a;
	SSG_DEVMEM bool operator()(int32_t x, int32_t y) const
	{ const int sx = a[x].score, sy = a[y].score; const uint64_t hx = a[x].hash, hy = a[y].hash; return (sx > sy) | ((sx == sy) & (hx < hy)); }
};

SSG_DEVFN int ssg_mark_primary_se(const ssg_mem_opt_t &opt, int n, ssg_alnreg_t *a, int64_t id, int32_t *z, int32_t *idx)
{	/* upstream mem_mark_primary_se + _core (ALT-free) */
	int i, k, tmp, zn = 0;
	if (n == 0) return 0;
	for (i = 0; i < n; ++i) { a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = ssg_hash64((uint64_t)(id + i)); }
	if (n <= 4) ssg_introsort(a, (long)n, ssg_reg_hash_lt());
	else {	/* (score, hash) keys are distinct (hash_64 is a bijection): sort 4-byte indices, then move each 88-byte record once */
		for (i = 0; i < n; ++i) idx[i] = i;
		ssg_reg_hash_idx_lt lt = { a };
		ssg_introsort(idx, (long)n, lt);
		for (int s0 = 0; s0 < n; ++s0) {
			if (idx[s0] < 0 || idx[s0] == s0) continue;
			const ssg_alnreg_t t = a[s0];
			int cur = s0;
			for (;;) {
				const int src = idx[cur];
				idx[cur] = -1 - src;
				if (src == s0) { a[cur] = t; break; }
				a[cur] = a[src];
				cur = src;
			}
		}
	}
	tmp = opt.a + opt.b;
	tmp = opt.o_del + opt.e_del > tmp ? opt.o_del + opt.e_del : tmp;
	tmp = opt.o_ins + opt.e_ins > tmp ? opt.o_ins + opt.e_ins : tmp;
	z[zn++] = 0;
	for (i = 1; i < n; ++i) {
		for (k = 0; k < zn; ++k) {
			int j = z[k];
			int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
			int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
			if (e_min > b_max) {
				int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
				if (e_min - b_max >= min_l * opt.mask_level) {
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == zn) z[zn++] = i;
		else a[i].secondary = z[k];
	}
	for (i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
	return n;
}

SSG_DEVFN int ssg_approx_mapq_se(const ssg_mem_opt_t &opt, const ssg_alnreg_t &a)
{	/* upstream mem_approx_mapq_se */
	int mapq, l, sub = a.sub ? a.sub : opt.min_seed_len * opt.a;
	double identity;
	sub = a.csub > sub ? a.csub : sub;
	if (sub >= a.score) return 0;
	l = a.qe - a.qb > a.re - a.rb ? a.qe - a.qb : (int)(a.re - a.rb);
	identity = 1. - (double)(l * opt.a - a.score) / (opt.a + opt.b) / l;
	if (a.score == 0) mapq = 0;
	else if (opt.mapQ_coef_len > 0) {
		double tmp;
		tmp = l < opt.mapQ_coef_len ? 1. : opt.mapQ_coef_fac / log((double)l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a.score - sub) / opt.a * tmp * tmp + .499);
	} else {
		mapq = (int)(30.0 * (1. - (double)sub / a.score) * log((double)a.seedcov) + .499);
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a.sub_n > 0) mapq -= (int)(4.343 * log((double)(a.sub_n + 1)) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a.frac_rep) + .499);
	return mapq;
}

typedef struct { uint64_t x, y; } ssg_pair64_t;
struct ssg_p128_lt { SSG_DEVMEM bool operator()(const ssg_pair64_t &a, const ssg_pair64_t &b) const { return (a.x < b.x) | ((a.x == b.x) & (a.y < b.y)); } }; /* branch-free: see ssg_chain_key_lt */

SSG_DEVFN int ssg_mem_pair(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const ssg_pestat_t *pes, ssg_alnreg_t *const a[2], int id,
                           int *sub, int *n_sub, int z[2], const int n_pri[2], ssg_pair64_t *v, ssg_pair64_t *u, int ucap, int *err)
{	/* upstream mem_pair */
	int r, i, k, y[4], ret, vn = 0, un = 0;
	const int64_t l_pac = ix.l_pac;
	for (r = 0; r < 2; ++r)
		for (i = 0; i < n_pri[r]; ++i) {
			ssg_pair64_t key;
			const ssg_alnreg_t &e = a[r][i];
			key.x = (uint64_t)(e.rb < l_pac ? e.rb : (l_pac << 1) - 1 - e.rb);
			key.x = (uint64_t)e.rid << 32 | (key.x - (uint64_t)ix.ctg_off[e.rid]);
			key.y = (uint64_t)e.score << 32 | (uint64_t)(i << 2) | (uint64_t)((e.rb >= l_pac) << 1) | (uint64_t)r;
			v[vn++] = key;
		}
	ssg_introsort(v, (long)vn, ssg_p128_lt());
	y[0] = y[1] = y[2] = y[3] = -1;
	for (i = 0; i < vn; ++i) {
		for (r = 0; r < 2; ++r) {
			int dir = r << 1 | (int)(v[i].y >> 1 & 1), which;
			if (pes[dir].failed) continue;
			which = r << 1 | (int)((v[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (k = y[which]; k >= 0; --k) {
				int64_t dist; int q; double ns;
				if ((int)(v[k].y & 3) != which) continue;
				dist = (int64_t)v[i].x - (int64_t)v[k].x;
				if (dist > pes[dir].high) break;
				if (dist < pes[dir].low) continue;
				ns = (dist - pes[dir].avg) / pes[dir].std;
				q = (int)((v[i].y >> 32) + (v[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt.a + .499);
				if (q < 0) q = 0;
				if (un >= ucap) { *err = 4; continue; }
				u[un].y = (uint64_t)k << 32 | (uint64_t)i;
				u[un].x = (uint64_t)q << 32 | (ssg_hash64(u[un].y ^ (uint64_t)(int64_t)(id << 8)) & 0xffffffffU);
				++un;
			}
		}
		y[v[i].y & 3] = i;
	}
	if (un) {
		int tmp = opt.a + opt.b;
		tmp = tmp > opt.o_del + opt.e_del ? tmp : opt.o_del + opt.e_del;
		tmp = tmp > opt.o_ins + opt.e_ins ? tmp : opt.o_ins + opt.e_ins;
		ssg_introsort(u, (long)un, ssg_p128_lt());
		i = (int)(u[un-1].y >> 32); k = (int)(u[un-1].y << 32 >> 32);
		z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
		z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
		ret = (int)(u[un-1].x >> 32);
		*sub = un > 1 ? (int)(u[un-2].x >> 32) : 0;
		for (i = un - 2, *n_sub = 0; i >= 0; --i)
			if (*sub - (int)(u[i].x >> 32) <= tmp) ++*n_sub;
	} else ret = 0, *sub = 0, *n_sub = 0;
	return ret;
}

#define SSG_RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))

/* XA requests for the regions shadowed by main record `main_reg` (upstream mem_gen_alt) */
SSG_DEVFN int ssg_emit_xa(const ssg_mem_opt_t &opt, const ssg_alnreg_t *a, int n, long abs0, int read, ssg_alnreq_t *req, int nreq, const int32_t *cnt)
{
	for (int i = 0; i < n; ++i) {
		int k = a[i].secondary_all;
		if (!(k >= 0 && a[i].score >= a[k].score * (double)opt.XA_drop_ratio)) continue;
		if (cnt[k] > opt.max_XA_hits_alt || cnt[k] > opt.max_XA_hits) continue;
		ssg_alnreq_t q; q.read = read; q.reg = (int32_t)(abs0 + i); q.kind = SSG_REQ_XA; q.owner = k; q.flag = 0; q.mapq = 0; q._pad0 = q._pad1 = 0;
		req[nreq++] = q;
	}
	return nreq;
}

/*
 * One lane per pair.  work: per-pair scratch of 2*(cap0+cap1) ssg_pair64_t for v[] plus ucap for u[],
 * zbuf: int32 per region slot.  req: per-read slices [req_off[r], req_off[r+1]); n_req out.
 * XA entries carry `owner` = region index (within the read) of the main record they belong to.
 */
/* upstream mem_sam_pe after mem_pair: the pairing / MAPQ decision and the list of records to generate.
 * o, subo, n_sub, z[] are mem_pair's results (o = 0 when it was not run). */
SSG_DEVFN void ssg_pair_decide(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const ssg_pestat_t *pes, const long p, ssg_alnreg_t *const a[2], const int an[2],
                               const int n_pri[2], const int o, int subo, const int n_sub, int z[2], int32_t *z0, const int64_t *reg_off, ssg_alnreq_t *const rq[2], int32_t *n_req)
{
	int extra_flag = 1, i, j;
	int nrq[2] = {0, 0};
	bool paired = false;
	int q_se[2] = {0, 0};
	if (n_pri[0] && n_pri[1] && o > 0) {
		int is_multi[2], q_pe, score_un;
		for (i = 0; i < 2; ++i) {
			for (j = 1; j < n_pri[i]; ++j) if (a[i][j].secondary < 0 && a[i][j].score >= opt.T) break;
			is_multi[i] = j < n_pri[i] ? 1 : 0;
		}
		if (!(is_multi[0] || is_multi[1])) {
			paired = true;
			score_un = a[0][0].score + a[1][0].score - opt.pen_unpaired;
			subo = subo > score_un ? subo : score_un;
			q_pe = SSG_RAW_MAPQ(o - subo, opt.a);
			if (n_sub > 0) q_pe -= (int)(4.343 * log((double)(n_sub + 1)) + .499);
			if (q_pe < 0) q_pe = 0;
			if (q_pe > 60) q_pe = 60;
			q_pe = (int)(q_pe * (1. - .5 * (a[0][0].frac_rep + a[1][0].frac_rep)) + .499);
			if (o > score_un) {
				ssg_alnreg_t *c[2] = { &a[0][z[0]], &a[1][z[1]] };
				for (i = 0; i < 2; ++i) {
					if (c[i]->secondary >= 0) { c[i]->sub = a[i][c[i]->secondary].score; c[i]->secondary = -2; }
					q_se[i] = ssg_approx_mapq_se(opt, *c[i]);
				}
				q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
				q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
				extra_flag |= 2;
				{ int c0 = SSG_RAW_MAPQ(c[0]->score - c[0]->csub, opt.a); q_se[0] = q_se[0] < c0 ? q_se[0] : c0; }
				{ int c1 = SSG_RAW_MAPQ(c[1]->score - c[1]->csub, opt.a); q_se[1] = q_se[1] < c1 ? q_se[1] : c1; }
			} else {
				z[0] = z[1] = 0;
				q_se[0] = ssg_approx_mapq_se(opt, a[0][0]);
				q_se[1] = ssg_approx_mapq_se(opt, a[1][0]);
			}
			for (i = 0; i < 2; ++i) {
				int k = a[i][z[i]].secondary_all;
				if (k >= 0 && k < n_pri[i]) {
					for (j = 0; j < an[i]; ++j) if (a[i][j].secondary_all == k || j == k) a[i][j].secondary_all = z[i];
					a[i][z[i]].secondary_all = -1;
				}
			}
			for (i = 0; i < 2; ++i) {
				ssg_alnreq_t q; q.read = (int32_t)(2*p + i); q.reg = (int32_t)(reg_off[2*p+i] + z[i]); q.kind = SSG_REQ_MAIN; q.owner = z[i];
				q.flag = (0x40 << i) | extra_flag; q.mapq = q_se[i]; q._pad0 = q._pad1 = 0;
				rq[i][nrq[i]++] = q;
			}
		}
	}
	if (!paired) { /* upstream no_pairing: */
		int hrid[2] = { -1, -1 };
		for (i = 0; i < 2; ++i) if (an[i] && a[i][0].score >= opt.T) hrid[i] = a[i][0].rid;
		if (hrid[0] == hrid[1] && hrid[0] >= 0) {
			int64_t dist; int d = ssg_infer_dir(ix.l_pac, a[0][0].rb, a[1][0].rb, &dist);
			if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
		}
		for (i = 0; i < 2; ++i) { /* upstream mem_reg2sam */
			int l = 0, k, mapq0 = 0;
			for (k = 0; k < an[i]; ++k) {
				const ssg_alnreg_t &pr = a[i][k];
				if (pr.score < opt.T) continue;
				if (pr.secondary >= 0) continue;
				ssg_alnreq_t q; q.read = (int32_t)(2*p + i); q.reg = (int32_t)(reg_off[2*p+i] + k); q.kind = SSG_REQ_MAIN; q.owner = k;
				q.flag = (i ? 0x81 : 0x41) | extra_flag; q._pad0 = q._pad1 = 0;
				q.mapq = pr.secondary < 0 ? ssg_approx_mapq_se(opt, pr) : 0;
				if (l) q.flag |= (opt.flag & SSG_F_NO_MULTI) ? 0x10000 : 0x800;   /* upstream mem_reg2sam */
				if (l && q.mapq > mapq0) q.mapq = mapq0;
				if (!l) mapq0 = q.mapq;
				rq[i][nrq[i]++] = q;
				++l;
			}
			if (l == 0) {
				ssg_alnreq_t q; q.read = (int32_t)(2*p + i); q.reg = -1; q.kind = SSG_REQ_MAIN; q.owner = -1;
				q.flag = (i ? 0x81 : 0x41) | extra_flag | 0x4; q.mapq = 0; q._pad0 = q._pad1 = 0;
				rq[i][nrq[i]++] = q;
			}
		}
	}
	for (i = 0; i < 2; ++i) { /* XA entries (upstream mem_gen_alt): count per primary, then emit for the main records */
		int32_t *cnt = z0; /* reuse */
		int nmain = nrq[i], tot = 0;
		for (j = 0; j < an[i]; ++j) cnt[j] = 0;
		for (j = 0; j < an[i]; ++j) {
			int k = a[i][j].secondary_all;
			if (k >= 0 && a[i][j].score >= a[i][k].score * (double)opt.XA_drop_ratio) { ++cnt[k]; ++tot; }
		}
		if (tot) {
			for (j = 0; j < an[i]; ++j) {
				int k = a[i][j].secondary_all;
				if (!(k >= 0 && a[i][j].score >= a[i][k].score * (double)opt.XA_drop_ratio)) continue;
				if (cnt[k] > opt.max_XA_hits_alt || cnt[k] > opt.max_XA_hits) continue;
				bool wanted = false;
				for (int m2 = 0; m2 < nmain; ++m2) if (rq[i][m2].owner == k && rq[i][m2].reg >= 0) wanted = true;
				if (!wanted) continue;
				ssg_alnreq_t q; q.read = (int32_t)(2*p + i); q.reg = (int32_t)(reg_off[2*p+i] + j); q.kind = SSG_REQ_XA; q.owner = k; q.flag = 0; q.mapq = 0; q._pad0 = q._pad1 = 0;
				rq[i][nrq[i]++] = q;
			}
		}
		n_req[2*p + i] = nrq[i];
	}
}

/* Single-end reads (upstream mem_process_seqs without MEM_F_PE: worker2 = mem_mark_primary_se(id = n_processed + i) + mem_reg2sam with no
 * mate).  One lane per read; the regions stay where stage 1 left them (reg_off = the read's seed offsets).  Requests as mem_reg2sam lists
 * the records: every primary-chain hit with score >= T, the first as the main line, the others supplementary (0x800, or 0x10000 with -M);
 * the XA candidates of the lines that will be printed. */
__global__ void __launch_bounds__(64) ssg_k_se_final(ssg_mem_opt_t opt, int n_reads, int64_t id0, const int64_t *reg_off, ssg_alnreg_t *regs, const int32_t *n_reg,
                               int32_t *zbuf, int32_t *ibuf, const int64_t *req_off, ssg_alnreq_t *req, int32_t *n_req)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	ssg_alnreg_t *a = regs + reg_off[r]; const int an = n_reg[r];
	int32_t *z0 = zbuf + reg_off[r];
	ssg_alnreq_t *rq = req + req_off[r];
	(void)ssg_mark_primary_se(opt, an, a, id0 + r, z0, ibuf + reg_off[r]);
	int nrq = 0, l = 0, k, j, mapq0 = 0;
	for (k = 0; k < an; ++k) {
		const ssg_alnreg_t &pr = a[k];
		if (pr.score < opt.T) continue;
		if (pr.secondary >= 0) continue;
		ssg_alnreq_t q; q.read = (int32_t)r; q.reg = (int32_t)(reg_off[r] + k); q.kind = SSG_REQ_MAIN; q.owner = k;
		q.flag = 0; q._pad0 = q._pad1 = 0;
		q.mapq = ssg_approx_mapq_se(opt, pr);
		if (l) q.flag |= (opt.flag & SSG_F_NO_MULTI) ? 0x10000 : 0x800;
		if (l && q.mapq > mapq0) q.mapq = mapq0;
		if (!l) mapq0 = q.mapq;
		rq[nrq++] = q;
		++l;
	}
	if (l == 0) {
		ssg_alnreq_t q; q.read = (int32_t)r; q.reg = -1; q.kind = SSG_REQ_MAIN; q.owner = -1;
		q.flag = 0x4; q.mapq = 0; q._pad0 = q._pad1 = 0;
		rq[nrq++] = q;
	}
	{	/* XA entries (upstream mem_gen_alt): count per primary, then emit for the main records */
		int32_t *cnt = z0;
		const int nmain = nrq; int tot = 0;
		for (j = 0; j < an; ++j) cnt[j] = 0;
		for (j = 0; j < an; ++j) {
			const int kk = a[j].secondary_all;
			if (kk >= 0 && a[j].score >= a[kk].score * (double)opt.XA_drop_ratio) { ++cnt[kk]; ++tot; }
		}
		if (tot) {
			for (j = 0; j < an; ++j) {
				const int kk = a[j].secondary_all;
				if (!(kk >= 0 && a[j].score >= a[kk].score * (double)opt.XA_drop_ratio)) continue;
				if (cnt[kk] > opt.max_XA_hits_alt || cnt[kk] > opt.max_XA_hits) continue;
				bool wanted = false;
				for (int m2 = 0; m2 < nmain; ++m2) if (rq[m2].owner == kk && rq[m2].reg >= 0) wanted = true;
				if (!wanted) continue;
				ssg_alnreq_t q; q.read = (int32_t)r; q.reg = (int32_t)(reg_off[r] + j); q.kind = SSG_REQ_XA; q.owner = kk; q.flag = 0; q.mapq = 0; q._pad0 = q._pad1 = 0;
				rq[nrq++] = q;
			}
		}
	}
	n_req[r] = nrq;
}

__global__ void ssg_k_se_caps(int n_reads, const int32_t *n_reg, int32_t *capq)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < n_reads) capq[r] = 2 * n_reg[r] + 2;
}

__global__ void __launch_bounds__(64) ssg_k_pair_final(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_pairs, int64_t id0,
                                 const int64_t *reg_off, ssg_alnreg_t *regs, const int32_t *n_reg, const int32_t *pair_batch, const ssg_pestat_t *pes_all,
                                 int32_t *zbuf, ssg_pair64_t *vbuf, ssg_pair64_t *ubuf, int ucap,
                                 const int64_t *req_off, ssg_alnreq_t *req, int32_t *n_req, int32_t *err, const int32_t *work_order, int pq_first,
                                 const int32_t *todo_list, const unsigned int *n_todo /* the pairs ssg_k_pair_final_lds left (NULL: pq_first .. n_pairs of work_order) */)
{
	long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
	ssg_pair64_t *u = ubuf + gt * (long)ucap;
	const long n_work = todo_list ? (long)*n_todo : (long)n_pairs;
	for (long pq = (todo_list ? 0 : pq_first) + gt; pq < n_work; pq += nt) {
		const long p = todo_list ? todo_list[pq] : work_order ? work_order[pq] : pq;   /* similar-cost pairs share a wave */
		const ssg_pestat_t *pes = pes_all + (long)pair_batch[p] * 4;
		const int64_t id = id0 + p;
		ssg_alnreg_t *a[2] = { regs + reg_off[2*p], regs + reg_off[2*p+1] };
		const int an[2] = { n_reg[2*p], n_reg[2*p+1] };
		int32_t *z0 = zbuf + reg_off[2*p];
		ssg_pair64_t *v = vbuf + reg_off[2*p];
		int n_pri[2], z[2] = {0, 0}, o = 0, subo = 0, n_sub = 0, myerr = 0;
		ssg_alnreq_t *rq[2] = { req + req_off[2*p], req + req_off[2*p+1] };
		n_pri[0] = ssg_mark_primary_se(opt, an[0], a[0], id << 1 | 0, z0, (int32_t*)v);
		n_pri[1] = ssg_mark_primary_se(opt, an[1], a[1], id << 1 | 1, z0, (int32_t*)v);
		if (n_pri[0] && n_pri[1] && !(opt.flag & SSG_F_NOPAIRING)) o = ssg_mem_pair(ix, opt, pes, a, (int)id, &subo, &n_sub, z, n_pri, v, u, ucap, &myerr);
		ssg_pair_decide(ix, opt, pes, p, a, an, n_pri, o, subo, n_sub, z, z0, reg_off, rq, n_req);
		if (myerr) err[p] = myerr;
	}
}

/* The same for the pairs with few candidate regions (nearly all of a batch), with the pair's state in the lane's slice of LDS: mem_mark_primary_se, mem_pair
 * and mem_sam_pe's decisions touch every field of every region record several times -- sorts included -- and on global memory each touch was a round trip
 * (round 5: 18 ms per million pairs at 0.6 % of the VALU rate, profiles/r05_pmc_sq.json).  Here a lane copies the regions of its two reads in (88 bytes each),
 * the same device functions run on the copies (sort scratch, pairing candidates and the z list lie in the slice too), and the records go back once.  Pairs with
 * more than SSG_PF_CAP regions in all are listed for ssg_k_pair_final. */
#ifndef SSG_PF_CAP
#define SSG_PF_CAP 6   /* 52 KB of LDS a wave: three waves a CU; 8 would pass the 64 KB of a static block */
#endif
#define SSG_PF_UCAP ((SSG_PF_CAP / 2) * (SSG_PF_CAP - SSG_PF_CAP / 2))     /* mem_pair lists a (hit of read 1, hit of read 2) combination at most once */
#define SSG_PF_SLICE ((SSG_PF_CAP * (int)sizeof(ssg_alnreg_t) + SSG_PF_CAP * 4 + SSG_PF_CAP * 16 + SSG_PF_UCAP * 16 + 8 + 15) / 16 * 16 + 8)   /* bytes per lane; an odd number of 8-byte words: the lanes' slices start in different banks */
__global__ void __launch_bounds__(64) ssg_k_pair_final_lds(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_pairs, int64_t id0,
                                 const int64_t *reg_off, ssg_alnreg_t *regs, const int32_t *n_reg, const int32_t *pair_batch, const ssg_pestat_t *pes_all,
                                 const int64_t *req_off, ssg_alnreq_t *req, int32_t *n_req, int32_t *err, const int32_t *work_order, int pq_first,
                                 int32_t *todo_list, unsigned int *n_todo)
{
	__shared__ uint64_t lds_[64 * SSG_PF_SLICE / 8];
	const long pq = pq_first + (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (pq >= n_pairs) return;
	const long p = work_order ? work_order[pq] : pq;
	const int an[2] = { n_reg[2*p], n_reg[2*p+1] };
	if (an[0] + an[1] > SSG_PF_CAP) { todo_list[atomicAdd(n_todo, 1u)] = (int32_t)p; return; }
	uint8_t *sl = (uint8_t*)lds_ + (size_t)threadIdx.x * SSG_PF_SLICE;
	ssg_alnreg_t *L = (ssg_alnreg_t*)sl;
	ssg_pair64_t *v = (ssg_pair64_t*)(sl + SSG_PF_CAP * sizeof(ssg_alnreg_t)), *u = v + SSG_PF_CAP;
	int32_t *z0 = (int32_t*)(u + SSG_PF_UCAP);
	ssg_alnreg_t *g[2] = { regs + reg_off[2*p], regs + reg_off[2*p+1] };
	ssg_alnreg_t *a[2] = { L, L + an[0] };
	for (int i = 0; i < 2; ++i) for (int k = 0; k < an[i]; ++k) a[i][k] = g[i][k];
	const ssg_pestat_t *pes = pes_all + (long)pair_batch[p] * 4;
	const int64_t id = id0 + p;
	int n_pri[2], z[2] = {0, 0}, o = 0, subo = 0, n_sub = 0, myerr = 0;
	ssg_alnreq_t *rq[2] = { req + req_off[2*p], req + req_off[2*p+1] };
	n_pri[0] = ssg_mark_primary_se(opt, an[0], a[0], id << 1 | 0, z0, (int32_t*)v);
	n_pri[1] = ssg_mark_primary_se(opt, an[1], a[1], id << 1 | 1, z0, (int32_t*)v);
	if (n_pri[0] && n_pri[1] && !(opt.flag & SSG_F_NOPAIRING)) o = ssg_mem_pair(ix, opt, pes, a, (int)id, &subo, &n_sub, z, n_pri, v, u, SSG_PF_UCAP, &myerr);
	ssg_pair_decide(ix, opt, pes, p, a, an, n_pri, o, subo, n_sub, z, z0, reg_off, rq, n_req);
	for (int i = 0; i < 2; ++i) for (int k = 0; k < an[i]; ++k) g[i][k] = a[i][k];
	if (myerr) err[p] = myerr;
}
#endif
