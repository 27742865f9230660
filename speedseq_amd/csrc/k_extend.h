/*
 * k_extend.h -- gfx950 kernels for seed extension.
 *
 *   ssg_k_extend_jobs   one wavefront per ksw_extend2 job (stage-level entry point; also the
 *                       kernel the SW micro-benchmark and rocprof roofline line are taken from).
 *   ssg_k_chain2aln_lane  one LANE per light read (a few chains with a few seeds): the scalar replay of upstream
 *                       mem_chain2aln + mem_sort_dedup_patch on the extension results of k_extlane.h; reads it cannot
 *                       finish go to a to-do list.
 *   ssg_k_chain2aln     one WAVEFRONT per read of that list (long chain lists of repeat-heavy reads, later-seed
 *                       extensions, patch alignments).  The per-read control flow is inherently sequential (every
 *                       seed is tested against the regions produced so far): it runs wave-uniform while region scans,
 *                       re-sorts and the DP rows of the rare in-place extension run lane-parallel.
 *
 * The first seed of every chain is extended ahead of time, one lane per extension (k_extlane.h); the 1-byte-per-base
 * window of a chain is decoded from the 2-bit .pac only when a later seed has to be extended here.
 */
#ifndef SSG_K_EXTEND_H
#define SSG_K_EXTEND_H
#include "k_sw.h"
#include "k_sdp.h"
#include "k_extlane.h"

#define SSG_TWIN_LDS 1024
#define SSG_TWIN_GLB 32768
#define SSG_WAVES_PER_WG 4
/* register budgets of the wave-per-item kernels are set by measurement (launch bounds below) */
#ifndef SSG_C2A_LKEYS
#define SSG_C2A_LKEYS 448   /* containment keys of a read's first regions in LDS, 24 bytes each: 10.5 KB a wave, 79 KB a workgroup with the rest -- two workgroups a CU, what 256 VGPRs allow anyway
                             * (144 until r06c2, laid over the re-sort's key area; repeat-heavy reads have hundreds of regions near one another and fetched the keys beyond from the HBM slab: a round trip per 64 regions and seed) */
#endif
#ifndef SSG_C2A_SCAN
#define SSG_C2A_SCAN 1   /* chunks of 64 region keys fetched per round trip of the containment scan (2 trips the backend's odd-aligned 64-bit reload bug at 168 VGPRs) */
#endif
#ifndef SSG_C2A_WAVES_PER_SIMD
#define SSG_C2A_WAVES_PER_SIMD 2   /* chain2aln: no spills at 2 (256 VGPRs); 18.0 ms against 18.9 at 3 (168 VGPRs, where the backend's odd-aligned 64-bit scratch reload error comes and goes with small changes) and 282 vs 266 ms a step at 4 (round 5) */
#endif
#ifndef SSG_SW_WAVES_PER_SIMD
#define SSG_SW_WAVES_PER_SIMD 3   /* mate rescue: 168 VGPRs; 91 ms against 106 at 4 waves (128 VGPRs) */
#endif

__global__ void __launch_bounds__(256) ssg_k_extend_jobs(ssg_mem_opt_t opt, int n_jobs, const ssg_ext_job_t *jobs, const uint8_t *qbuf, const uint8_t *tbuf,
                                  ssg_ext_res_t *res, unsigned long long *cells)
{
	long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (wid >= n_jobs) return;
	ssg_ext_job_t jb = jobs[wid];
	ssg_seqv_t q = { qbuf + jb.qoff, 1 }, t = { tbuf + jb.toff, 1 };
	unsigned long long nc = 0;
	ssg_ext_res_t r = wv_extend2_any<true>(opt, jb.qlen, q, jb.tlen, t, jb.w, jb.end_bonus, jb.zdrop, jb.h0, &nc);
	if (wv_lane() == 0) { res[wid] = r; if (cells) atomicAdd(cells, nc); }
}

SSG_DEVFN int ssg_cal_max_gap(const ssg_mem_opt_t &opt, int qlen)
{
	int l_del = (int)((double)(qlen * opt.a - opt.o_del) / opt.e_del + 1.);
	int l_ins = (int)((double)(qlen * opt.a - opt.o_ins) / opt.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < opt.w << 1 ? l : opt.w << 1;
}

/* cooperative decode of reference [beg,end) (doubled coordinates) into dst */
SSG_DEVFN void wv_fetch_ref(const ssg_index_view_t &ix, int64_t beg, int64_t end, uint8_t *dst)
{
	ssg_wave_memsync();
	for (int64_t k = beg + wv_lane(); k < end; k += 64) dst[k - beg] = (uint8_t)ssg_ref_base(ix, k);
	ssg_wave_memsync();
}

struct ssg_u64_lt { SSG_DEVMEM bool operator()(uint64_t a, uint64_t b) const { return a < b; } };
struct ssg_reg_re_lt { SSG_DEVMEM bool operator()(const ssg_alnreg_t &a, const ssg_alnreg_t &b) const { return a.re < b.re; } };
struct ssg_reg_sc_lt {
	SSG_DEVMEM bool operator()(const ssg_alnreg_t &a, const ssg_alnreg_t &b) const
	{ return (a.score > b.score) | ((a.score == b.score) & ((a.rb < b.rb) | ((a.rb == b.rb) & (a.qb < b.qb)))); } /* branch-free: see ssg_chain_key_lt */
};


/* score of the banded global alignment of query[qb,qe) vs reference [rb,re) as upstream
 * bwa_gen_cigar2 computes it with n_cigar == NULL (both reversed when on the reverse strand) */
template <bool WIDE>
SSG_DEVFN int wv_gen_score(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, int w_, int l_query, const uint8_t *query,
                           int64_t rb, int64_t re, uint8_t *tbuf, int *ok, unsigned long long *cells)
{
	*ok = 0;
	if (l_query <= 0 || rb >= re || (rb < ix.l_pac && re > ix.l_pac)) return 0;
	if (rb < 0 || re > ix.l_pac << 1) return 0;
	int rlen = (int)(re - rb);
	wv_fetch_ref(ix, rb, re, tbuf);
	*ok = 1;
	const bool rev = rb >= ix.l_pac;
	ssg_seqv_t q = { rev ? query + l_query - 1 : query, rev ? -1 : 1 };
	ssg_seqv_t t = { rev ? tbuf + rlen - 1 : tbuf, rev ? -1 : 1 };
	if (l_query == rlen && w_ == 0) {
		int sc = 0;
		for (int i = wv_lane(); i < l_query; i += 64) sc += opt.mat[sq_at(t, i) * 5 + sq_at(q, i)];
		return wv_sum(sc);
	}
	int max_ins = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
	int max_del = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
	int max_gap = max_ins > max_del ? max_ins : max_del;
	max_gap = max_gap > 1 ? max_gap : 1;
	int w = (max_gap + iabs(rlen - l_query) + 1) >> 1;
	w = w < w_ ? w : w_;
	int min_w = iabs(rlen - l_query) + 3;
	w = w > min_w ? w : min_w;
	return wv_global2_any<WIDE>(opt, l_query, q, rlen, t, w, 0, cells);
}

template <bool WIDE>
SSG_DEVFN int wv_patch_reg(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const uint8_t *query, const ssg_alnreg_t &a, const ssg_alnreg_t &b,
                           int *_w, uint8_t *tbuf, int tcap, int *err, unsigned long long *cells)
{	/* upstream mem_patch_reg */
	int w, score, q_s, r_s, ok; double r;
	if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
	w = (int)((a.re - b.rb) - (a.qe - b.qb));
	w = w > 0 ? w : -w;
	r = (double)(a.re - b.rb) / (b.re - a.rb) - (double)(a.qe - b.qb) / (b.qe - a.qb);
	r = r > 0. ? r : -r;
	if (a.re < b.rb || a.qe < b.qb) { if (w > opt.w << 1 || r >= SSG_PATCH_MAX_R_BW) return 0; }
	else if (w > opt.w << 2 || r >= SSG_PATCH_MAX_R_BW * 2) return 0;
	w += a.w + b.w;
	w = w < opt.w << 2 ? w : opt.w << 2;
	if (b.re - a.rb > tcap) { *err = 1; return 0; }
	score = wv_gen_score<WIDE>(ix, opt, w, b.qe - a.qb, query + a.qb, a.rb, b.re, tbuf, &ok, cells);
	if (!ok) score = 0; /* upstream leaves score unset when bwa_gen_cigar2 bails out; unreachable for same-strand regs */
	q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);
	r_s = (int)((double)(b.re - a.rb) / ((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < SSG_PATCH_MIN_SC_RATIO) return 0;
	*_w = w;
	return score;
}

/* upstream mem_sort_dedup_patch (wave-uniform; `patch` enables mem_patch_reg as in mem_align1_core) */
template <bool WIDE>
SSG_DEVFN_COLD int wv_sort_dedup_patch(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const uint8_t *query, int patch, int n, ssg_alnreg_t *a,
                                  uint8_t *tbuf, int tcap, int *err, unsigned long long *cells)
{
	int m, i, j;
	if (n <= 1) return n;
	SSG_LANE0(ssg_introsort(a, (long)n, ssg_reg_re_lt()); for (int t = 0; t < n; ++t) a[t].n_comp = 1);
	for (i = 1; i < n; ++i) {
		ssg_alnreg_t *p = &a[i];
		if (p->rid != a[i-1].rid || p->rb >= a[i-1].re + opt.max_chain_gap) continue;
		for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt.max_chain_gap; --j) {
			ssg_alnreg_t *q = &a[j];
			int64_t or_, oq, mr, mq; int score, w;
			if (q->qe == q->qb) continue;
			or_ = q->re - p->rb;
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (or_ > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq) {
				if (p->score < q->score) { SSG_LANE0(p->qe = p->qb); break; }
				else SSG_LANE0(q->qe = q->qb);
			} else if (patch && q->rb < p->rb && (score = wv_patch_reg<WIDE>(ix, opt, query, *q, *p, &w, tbuf, tcap, err, cells)) > 0) {
				SSG_LANE0(
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub ? p->sub : q->sub;
				p->csub = p->csub > q->csub ? p->csub : q->csub;
				p->qb = q->qb; p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe);
			}
		}
	}
	m = 0;
	SSG_LANE0(
		int mm = 0, t;
		for (t = 0; t < n; ++t)
			if (a[t].qe > a[t].qb) { if (mm != t) a[mm++] = a[t]; else ++mm; }
		int nn = mm;
		ssg_introsort(a, (long)nn, ssg_reg_sc_lt());
		for (t = 1; t < nn; ++t)
			if (a[t].score == a[t-1].score && a[t].rb == a[t-1].rb && a[t].qb == a[t-1].qb) a[t].qe = a[t].qb;
		for (t = 1, mm = 1; t < nn; ++t)
			if (a[t].qe > a[t].qb) { if (mm != t) a[mm++] = a[t]; else ++mm; }
		m = nn < 1 ? nn : mm);
	return wv_bcast(m, 0);
}

#define SSG_MAX_BAND_TRY 2

/* upstream mem_chain2aln: is seed s already covered by region (rb,re,qb,qe) of band w, first seed length seedlen0? */
SSG_DEVFN int ssg_seed_in_region(const ssg_mem_opt_t &opt, const ssg_seed_t &s, int l_query, int64_t prb, int64_t pre, int pqb, int pqe, int pw, int psl)
{
	if (s.rbeg < prb || s.rbeg + s.len > pre || s.qbeg < pqb || s.qbeg + s.len > pqe) return 0;
	if (s.len - psl > .1 * l_query) return 0;
	int64_t rd; int qd, w, max_gap;
	qd = s.qbeg - pqb; rd = s.rbeg - prb;
	max_gap = ssg_cal_max_gap(opt, qd < rd ? qd : (int)rd);
	w = max_gap < pw ? max_gap : pw;
	if (qd - rd < w && rd - qd < w) return 1;
	qd = pqe - (s.qbeg + s.len); rd = pre - (s.rbeg + s.len);
	max_gap = ssg_cal_max_gap(opt, qd < rd ? qd : (int)rd);
	w = max_gap < pw ? max_gap : pw;
	return qd - rd < w && rd - qd < w;
}

/*
 * One wavefront per read.  Per-read slices start at seed_off[r]: chains[], order[] (surviving chain
 * ids), srt[] (u64 work array), regs[] (capacity = #seeds of the read).  n_reg[r] receives the
 * number of regions left after mem_sort_dedup_patch; err[r] != 0 flags a window overflow.
 */
template <bool WIDE>
SSG_DEVFN void wv_chain2aln_read(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const long r, const uint8_t *seq, const int64_t *read_off,
                                const int64_t *seed_off, const ssg_seed_t *seeds, const ssg_chain_t *chains, const int32_t *order,
                                const int32_t *chain_seeds, const int32_t *n_chain, uint64_t *srt_all, ssg_alnreg_t *regs, int32_t *n_reg,
                                uint8_t *tlds_w, uint8_t *tg, int32_t *err, unsigned long long *cells, unsigned long long *ph,
                                ssg_sdp_small_t *sdp_lds, ssg_sdp_big_t *sdp_big, ssg_alnreg_t *sdp_tmp,
                                const int64_t *chain_off, const ssg_xjob_t *xjobs, const ssg_xres_t *xres_l, const ssg_xres_t *xres_r, uint8_t *sdp_fixed,
                                uint16_t *hq /* LDS, SSG_SDP_BIG entries: (rb >> 10) of the read's regions so far */, uint64_t *lk /* LDS, 3 x SSG_C2A_LKEYS words */)
{
	const uint8_t *query = seq + read_off[r];
	const int l_query = (int)(read_off[r+1] - read_off[r]);
	const long s0 = seed_off[r];
	uint64_t *srt = srt_all + s0;
	ssg_alnreg_t *av = regs + s0;
	const int64_t l_pac = ix.l_pac;
	int av_n = 0, myerr = 0;
	int hq_ok = 1;   /* every region so far spans at most 1024 reference bases: one containing a seed starts in the seed's 1024-base bin or the one before */
	unsigned long long nc = 0;
	const int nch = n_chain[r];
	ssg_sdp_key_t *const ck = sdp_big->key;   /* (rb, re, qb, qe, w -> .score, seedlen0 -> .rid) of av[]: what the containment test reads */
	/* the first SSG_C2A_LKEYS of them also in LDS: 8 + 8 + 8 bytes */
	int64_t *const lk_rb = (int64_t*)lk, *const lk_re = lk_rb + SSG_C2A_LKEYS; uint64_t *const lk_m = (uint64_t*)(lk_re + SSG_C2A_LKEYS);
	unsigned long long t0 = 0, t1;
#define SSG_PH(x) do { if (SSG_TUNING && ph) { t1 = ssg_clock(); ph[x] += t1 - t0; t0 = t1; } } while (0)
	if (SSG_TUNING && ph) { t0 = ssg_clock(); }
	/* the next chain's record and the results of its first seed's extensions are fetched while this chain is worked on: a chain is a string of dependent round trips
	 * (record -> seed -> results), a read of the wave kernel has hundreds of chains, and the three loads here were a third of them */
	const long gid0 = (long)chain_off[r];
	ssg_xjob_t xj_n; ssg_xres_t xl_n, xr_n;
	if (nch > 0) { xj_n = xjobs[gid0]; xl_n = xres_l[gid0]; xr_n = xres_r[gid0]; }
	for (int ci = 0; ci < nch; ++ci) {
		/* the chain's record (ssg_k_ext_prep): window, best seed, seed count, contig, frac_rep -- no walk through chains[] / order[] */
		const long gid = gid0 + ci;
		const ssg_xjob_t xj = xj_n; const ssg_xres_t xl_c = xl_n, xr_c = xr_n;
		if (ci + 1 < nch) { xj_n = xjobs[gid + 1]; xl_n = xres_l[gid + 1]; xr_n = xres_r[gid + 1]; }
		struct { int n, rid; float frac_rep; } c = { xj.cn, xj.rid, xj.frac_rep };
		const int32_t *cs = chain_seeds + xj.first_seed;
		int i, k, max_off[2], aw[2];
		int64_t rmax[2], tmp;
		if (c.n == 0) continue;
		/* window and first seed's extensions were prepared by ssg_k_ext_prep / ssg_k_ext_lane (k_extlane.h) */
		rmax[0] = xj.rmax0; rmax[1] = xj.rmax1;
		const int span = (int)(rmax[1] - rmax[0]);
		uint8_t *rseq = span <= SSG_TWIN_LDS ? tlds_w : tg;
		if (xj.flag) { myerr = 1; continue; }
		int fetched = 0;   /* the 1-byte-per-base window is only needed when a later seed of the chain is extended here */
		/* chains of 2..64 seeds (a read of this kernel has hundreds): lane t keeps seed t, its rank in upstream's order (score, index) and its `srt[] = 0' mark in
		 * registers; the seed of a rank comes over by v_readlane and the walk along the later seeds is one test by their lanes -- through srt[] -> chain_seeds[] ->
		 * seeds[] in global memory every seed after the first cost three dependent round trips, and every step of that walk three more */
		const bool in_regs = c.n >= 2 && c.n <= 64;
		ssg_seed_t ms; ms.rbeg = 0; ms.qbeg = ms.len = ms.score = 0; ms.next = -1;
		int my_rnk = -1; bool my_alive = false;
		if (c.n == 1) { /* nothing to order; the only seed is in the record */ }
		else if (in_regs) { /* seeds by (score, index): distinct keys, rank = number of smaller keys, one lane per seed */
			const int t = wv_lane();
			if (t < c.n) ms = seeds[cs[t]];
			const uint64_t key = t < c.n ? ((uint64_t)ms.score << 32 | (uint64_t)t) : ~0ull;
			int rnk = 0;
			for (int j = 0; j < c.n; ++j) rnk += (uint64_t)wv_get64((long long)key, j) < key;
			if (t < c.n) { my_rnk = rnk; my_alive = true; }
		} else {
			SSG_LANE0(for (int t = 0; t < c.n; ++t) srt[t] = (uint64_t)seeds[cs[t]].score << 32 | (uint64_t)t;
			          ssg_introsort(srt, (long)c.n, ssg_u64_lt()));
		}
		SSG_PH(0);
		for (k = c.n - 1; k >= 0; --k) {
			ssg_seed_t s;
			int own = 0;   /* the lane that holds the seed of rank k */
			if (in_regs) own = (int)__builtin_ctzll(wv_ballot(my_rnk == k) | 1ull << 63);
			if (k == c.n - 1) { s.rbeg = xj.rbeg; s.qbeg = xj.qbeg; s.len = s.score = xj.len; s.next = -1; }   /* the best seed travels in the record */
			else if (in_regs) { s.rbeg = (int64_t)wv_get64((long long)ms.rbeg, own); s.qbeg = wv_get(ms.qbeg, own); s.len = wv_get(ms.len, own); s.score = wv_get(ms.score, own); s.next = -1; }
			else s = seeds[cs[(uint32_t)srt[k]]];
			{	/* is the seed contained in an earlier region?  Compact keys of the regions (ck[]), SSG_C2A_SCAN x 64 regions per round
				 * trip; the scalar loop's first hit decides */
				int hit = av_n;
				if (av_n <= SSG_SDP_BIG && hq_ok) {
					/* The scan is quadratic in a read's regions and nearly always finds nothing (repeat copies lie elsewhere): 64 regions' bins from LDS per
					 * step, the keys themselves (HBM slab: 1900 cycles a step, a quarter of this kernel) only for a region in the seed's bin */
					const uint16_t hs = (uint16_t)(s.rbeg >> 10), hs1 = (uint16_t)(hs - 1);
					for (int i0 = 0; i0 < av_n && hit == av_n; i0 += 64) {
						if (SSG_TUNING && ph) ++ph[5];
						const int ii = i0 + wv_lane();
						const uint16_t hv = ii < av_n ? hq[ii] : (uint16_t)(hs + 2);
						const bool cand = hv == hs || hv == hs1;
						if (!wv_ballot(cand)) continue;
						int h = 0;
						if (cand) {
							ssg_sdp_key_t kk;
							if (ii < SSG_C2A_LKEYS) { const uint64_t m = lk_m[ii]; kk.rb = lk_rb[ii]; kk.re = lk_re[ii]; kk.qb = (int)(m & 0xffff); kk.qe = (int)(m >> 16 & 0xffff); kk.score = (int)(m >> 32 & 0xffff); kk.rid = (int)(m >> 48); }
							else kk = ck[ii];
							h = ssg_seed_in_region(opt, s, l_query, kk.rb, kk.re, kk.qb, kk.qe, kk.score, kk.rid);
						}
						const unsigned long long bal = wv_ballot(h);
						if (bal) hit = i0 + (int)__builtin_ctzll(bal);
					}
				} else if (av_n <= SSG_SDP_BIG) {
					for (int i0 = 0; i0 < av_n && hit == av_n; i0 += 64 * SSG_C2A_SCAN) {
						if (SSG_TUNING && ph) ++ph[5];
						ssg_sdp_key_t kk[SSG_C2A_SCAN];
						SSG_UNROLL for (int u = 0; u < SSG_C2A_SCAN; ++u) {
							const int ii = i0 + u * 64 + wv_lane();
							if (ii < av_n) {
								if (ii < SSG_C2A_LKEYS) { const uint64_t m = lk_m[ii]; kk[u].rb = lk_rb[ii]; kk[u].re = lk_re[ii]; kk[u].qb = (int)(m & 0xffff); kk[u].qe = (int)(m >> 16 & 0xffff); kk[u].score = (int)(m >> 32 & 0xffff); kk[u].rid = (int)(m >> 48); }
								else kk[u] = ck[ii];
							}
						}
						SSG_UNROLL for (int u = 0; u < SSG_C2A_SCAN; ++u) {
							const int ii = i0 + u * 64 + wv_lane();
							const int h = hit == av_n && ii < av_n && ssg_seed_in_region(opt, s, l_query, kk[u].rb, kk[u].re, kk[u].qb, kk[u].qe, kk[u].score, kk[u].rid);
							const unsigned long long bal = wv_ballot(h);
							if (bal && hit == av_n) hit = i0 + u * 64 + (int)__builtin_ctzll(bal);
						}
					}
				} else {
					for (int i0 = 0; i0 < av_n && hit == av_n; i0 += 64) {
						const int ii = i0 + wv_lane();
						int h = 0;
						if (ii < av_n) { const ssg_alnreg_t *p = &av[ii]; h = ssg_seed_in_region(opt, s, l_query, p->rb, p->re, p->qb, p->qe, p->w, p->seedlen0); }
						const unsigned long long bal = wv_ballot(h);
						if (bal) hit = i0 + (int)__builtin_ctzll(bal);
					}
				}
				i = hit;
			}
			if (i < av_n && in_regs) {   /* upstream's walk over the seeds of higher rank leaves early at the first one that overlaps s off its diagonal: only whether one does matters */
				bool brk = false;
				if (my_rnk > k && my_alive && !(ms.len < s.len * .95))
					brk = (s.qbeg <= ms.qbeg && s.qbeg + s.len - ms.qbeg >= s.len >> 2 && ms.qbeg - s.qbeg != ms.rbeg - s.rbeg)
					   || (ms.qbeg <= s.qbeg && ms.qbeg + ms.len - s.qbeg >= s.len >> 2 && s.qbeg - ms.qbeg != s.rbeg - ms.rbeg);
				if (!wv_ballot(brk)) { if (wv_lane() == own) my_alive = false; SSG_PH(1); continue; }
			} else if (i < av_n) {
				for (i = k + 1; i < c.n; ++i) {
					if (srt[i] == 0) continue;
					const ssg_seed_t t = seeds[cs[(uint32_t)srt[i]]];
					if (t.len < s.len * .95) continue;
					if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
					if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
				}
				if (i == c.n) { if (c.n > 1) { SSG_LANE0(srt[k] = 0); } SSG_PH(1); continue; }
			}
			SSG_PH(1);
			ssg_alnreg_t a;
			a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.alt_sc = a.csub = a.sub_n = a.seedcov = a.secondary = a.secondary_all = a.n_comp = 0; a.hash = 0;
			a.w = aw[0] = aw[1] = opt.w;
			a.score = a.truesc = -1;
			a.rid = c.rid;
			const bool ahead = k == c.n - 1;   /* == xj.seed_t: extended ahead of time, one lane per extension */
			if (!ahead && !fetched) { wv_fetch_ref(ix, rmax[0], rmax[1], rseq); fetched = 1; }
			if (s.qbeg) { /* left extension: both sequences walked backwards */
				ssg_ext_res_t x; x.score = -1;
				tmp = s.rbeg - rmax[0];
				ssg_seqv_t qs = { query + s.qbeg - 1, -1 }, rs = { rseq + tmp - 1, -1 };
				if (ahead) {
					const ssg_xres_t o = xl_c;
					x.score = o.score; x.qle = o.qle; x.tle = o.tle; x.gtle = o.gtle; x.gscore = o.gscore; x.max_off = o.max_off;
					aw[0] = o.aw; a.score = o.score; max_off[0] = o.max_off;
				} else
				for (i = 0; i < SSG_MAX_BAND_TRY; ++i) {
					int prev = a.score;
					aw[0] = opt.w << i;
					x = wv_extend2_any<WIDE>(opt, s.qbeg, qs, (int)tmp, rs, aw[0], opt.pen_clip5, opt.zdrop, s.len * opt.a, &nc);
					a.score = x.score; max_off[0] = x.max_off;
					if (a.score == prev || max_off[0] < (aw[0] >> 1) + (aw[0] >> 2)) break;
				}
				if (x.gscore <= 0 || x.gscore <= a.score - opt.pen_clip5) { a.qb = s.qbeg - x.qle; a.rb = s.rbeg - x.tle; a.truesc = a.score; }
				else { a.qb = 0; a.rb = s.rbeg - x.gtle; a.truesc = x.gscore; }
			} else { a.score = a.truesc = s.len * opt.a; a.qb = 0; a.rb = s.rbeg; }
			if (s.qbeg + s.len != l_query) { /* right extension */
				ssg_ext_res_t x; x.score = -1;
				int qe = s.qbeg + s.len, sc0 = a.score;
				int re = (int)(s.rbeg + s.len - rmax[0]);
				ssg_seqv_t qs = { query + qe, 1 }, rs = { rseq + re, 1 };
				if (ahead) {
					const ssg_xres_t o = xr_c;
					x.score = o.score; x.qle = o.qle; x.tle = o.tle; x.gtle = o.gtle; x.gscore = o.gscore; x.max_off = o.max_off;
					aw[1] = o.aw; a.score = o.score; max_off[1] = o.max_off;
				} else
				for (i = 0; i < SSG_MAX_BAND_TRY; ++i) {
					int prev = a.score;
					aw[1] = opt.w << i;
					x = wv_extend2_any<WIDE>(opt, l_query - qe, qs, (int)(rmax[1] - rmax[0] - re), rs, aw[1], opt.pen_clip3, opt.zdrop, sc0, &nc);
					a.score = x.score; max_off[1] = x.max_off;
					if (a.score == prev || max_off[1] < (aw[1] >> 1) + (aw[1] >> 2)) break;
				}
				if (x.gscore <= 0 || x.gscore <= a.score - opt.pen_clip3) { a.qe = qe + x.qle; a.re = rmax[0] + re + x.tle; a.truesc += a.score - sc0; }
				else { a.qe = l_query; a.re = rmax[0] + re + x.gtle; a.truesc += x.gscore - sc0; }
			} else { a.qe = l_query; a.re = s.rbeg + s.len; }
			if (c.n == 1) a.seedcov = s.qbeg >= a.qb && s.qbeg + s.len <= a.qe && s.rbeg >= a.rb && s.rbeg + s.len <= a.re ? s.len : 0;
			else if (in_regs) { const int cov = my_rnk >= 0 && ms.qbeg >= a.qb && ms.qbeg + ms.len <= a.qe && ms.rbeg >= a.rb && ms.rbeg + ms.len <= a.re ? ms.len : 0; a.seedcov = wv_sum(cov); }
			else {   /* a lane per seed (the wave walking the chain's seeds together paid two dependent round trips a seed: index, then seed) */
				int cov = 0;
				for (int i0 = 0; i0 < c.n; i0 += 64) {
					const int i2 = i0 + wv_lane();
					if (i2 < c.n) { const ssg_seed_t t = seeds[cs[i2]]; if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len; }
				}
				a.seedcov = wv_sum(cov);
			}
			a.w = aw[0] > aw[1] ? aw[0] : aw[1];
			a.seedlen0 = s.len;
			a.frac_rep = c.frac_rep;
			SSG_PH(7);
			SSG_LANE0(av[av_n] = a;
			          if (av_n < SSG_C2A_LKEYS) { lk_rb[av_n] = a.rb; lk_re[av_n] = a.re; lk_m[av_n] = (uint64_t)(uint16_t)a.qb | (uint64_t)(uint16_t)a.qe << 16 | (uint64_t)(uint16_t)a.w << 32 | (uint64_t)(uint16_t)a.seedlen0 << 48; }
			          else if (av_n < SSG_SDP_BIG) { ssg_sdp_key_t ka; ka.re = a.re; ka.rb = a.rb; ka.qb = a.qb; ka.qe = a.qe; ka.score = a.w; ka.rid = a.seedlen0; ck[av_n] = ka; }
			          if (av_n < SSG_SDP_BIG) hq[av_n] = (uint16_t)(a.rb >> 10););
			if (a.re - a.rb > 1024 || a.rb < 0) hq_ok = 0;
			++av_n;
			SSG_PH(6);
		}
	}
	{	/* mem_sort_dedup_patch: on compact keys unless a pair of regions has to be globally aligned (patched) */
		int m = -1;
		if (av_n <= SSG_SDP_SMALL) m = wv_sort_dedup_fast(opt, av_n, av, sdp_tmp, sdp_lds->key, sdp_lds->skey, sdp_lds->idx, sdp_lds->idx2, l_pac);
		else if (av_n <= SSG_SDP_BIG) m = wv_sort_dedup_fast(opt, av_n, av, sdp_tmp, sdp_big->key, sdp_big->skey, sdp_big->idx, sdp_big->idx2, l_pac);
		av_n = m >= 0 ? m : wv_sort_dedup_patch<WIDE>(ix, opt, query, 1, av_n, av, tg, SSG_TWIN_GLB, &myerr, &nc);
		/* no pair of regions reached mem_patch_reg's alignment: the list is the output of the plain redundancy scan, a fixed point of it
		 * (mate rescue's first re-sort of this list can be the incremental one, k_sdp.h wv_sort_dedup_incr) */
		if (wv_lane() == 0 && sdp_fixed) sdp_fixed[r] = m >= 0;
	}
	SSG_PH(3);
#undef SSG_PH
	if (wv_lane() == 0) { n_reg[r] = av_n; err[r] = myerr; }
	*cells += nc;
}

/*
 * Light reads (a few chains, a few seeds each -- 98 % of a batch): one LANE per read.  With a wavefront per read the
 * 2 M reads of a batch file through ~3000 resident waves, each read a chain of ~40 dependent memory round trips; one lane
 * per read keeps 64 of those chains in flight per wave.  Everything is the scalar form of wv_chain2aln_read: the first
 * seed of each chain takes its extension from ssg_k_ext_lane's results; a read that would need anything only the wave
 * path has (an extension of a later seed, a patch alignment in mem_sort_dedup_patch, long lists) is left untouched
 * and flagged for the wave kernel.
 */
#ifndef SSG_C2A_LANE_CHAINS
#define SSG_C2A_LANE_CHAINS 6
#endif
#ifndef SSG_C2A_LANE_SEEDS
#define SSG_C2A_LANE_SEEDS 8
#endif
__global__ void __launch_bounds__(64) ssg_k_chain2aln_lane(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const int64_t *read_off, const int64_t *seed_off,
                                const ssg_seed_t *seeds, const int32_t *chain_seeds, const int32_t *n_chain, ssg_alnreg_t *regs, int32_t *n_reg,
                                int32_t *err, const int64_t *chain_off, const ssg_xjob_t *xjobs, const ssg_xres_t *xres_l, const ssg_xres_t *xres_r,
                                const int32_t *work_order, int32_t *todo_list, unsigned int *n_todo /* out: reads the wave kernel has to do, roughly heaviest first */,
                                uint8_t *sdp_fixed /* out: the list is a fixed point of the redundancy scan (always, here: a patch candidate sends the read to the wave kernel) */)
{
	const long g_ = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g_ >= n_reads) return;
	const long r = work_order ? work_order[g_] : g_;
#define SSG_C2A_TODO() do { todo_list[atomicAdd(n_todo, 1u)] = (int32_t)r; return; } while (0)
	const int nch = n_chain[r];
	if (nch > SSG_C2A_LANE_CHAINS) SSG_C2A_TODO();
	const int l_query = (int)(read_off[r+1] - read_off[r]);
	ssg_alnreg_t *av = regs + seed_off[r];
	const int64_t l_pac = ix.l_pac;
	int av_n = 0;
	for (int ci = 0; ci < nch; ++ci) {
		const long gid = (long)chain_off[r] + ci;
		const ssg_xjob_t xj = xjobs[gid];
		const int cn = xj.cn;
		if (cn == 0) continue;
		if (xj.flag || cn > SSG_C2A_LANE_SEEDS) SSG_C2A_TODO();
		const int32_t *cs = chain_seeds + xj.first_seed;
		uint64_t srt[SSG_C2A_LANE_SEEDS];
		if (cn > 1) { /* seeds by (score, index), ascending: distinct keys, insertion sort */
			for (int t = 0; t < cn; ++t) {
				const uint64_t key = (uint64_t)seeds[cs[t]].score << 32 | (uint64_t)t;
				int u = t;
				SSG_UNROLL for (int v = SSG_C2A_LANE_SEEDS - 1; v > 0; --v) if (v <= t && u == v && srt[v-1] > key) { srt[v] = srt[v-1]; u = v - 1; }
				SSG_UNROLL for (int v = 0; v < SSG_C2A_LANE_SEEDS; ++v) if (v == u) srt[v] = key;
			}
		}
		for (int k = cn - 1; k >= 0; --k) {
			ssg_seed_t s;
			uint64_t sk = 0;
			SSG_UNROLL for (int v = 0; v < SSG_C2A_LANE_SEEDS; ++v) if (v == k) sk = srt[v];
			if (k == cn - 1) { s.rbeg = xj.rbeg; s.qbeg = xj.qbeg; s.len = s.score = xj.len; s.next = -1; }
			else { if (sk == 0) continue; s = seeds[cs[(uint32_t)sk]]; }
			int i;
			for (i = 0; i < av_n; ++i) { const ssg_alnreg_t *p = &av[i]; if (ssg_seed_in_region(opt, s, l_query, p->rb, p->re, p->qb, p->qe, p->w, p->seedlen0)) break; }
			if (i < av_n) {
				for (i = k + 1; i < cn; ++i) {
					uint64_t ski = 0;
					SSG_UNROLL for (int v = 0; v < SSG_C2A_LANE_SEEDS; ++v) if (v == i) ski = srt[v];
					if (ski == 0) continue;
					ssg_seed_t t;
					if (i == cn - 1) { t.rbeg = xj.rbeg; t.qbeg = xj.qbeg; t.len = t.score = xj.len; t.next = -1; } else t = seeds[cs[(uint32_t)ski]];
					if (t.len < s.len * .95) continue;
					if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
					if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
				}
				if (i == cn) { SSG_UNROLL for (int v = 0; v < SSG_C2A_LANE_SEEDS; ++v) if (v == k) srt[v] = 0; continue; }
			}
			if (k != cn - 1) SSG_C2A_TODO();   /* a later seed has to be extended: wave kernel (nothing has been written that it does not rewrite) */
			ssg_alnreg_t a;
			a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.alt_sc = a.csub = a.sub_n = a.seedcov = a.secondary = a.secondary_all = a.n_comp = 0; a.hash = 0;
			int aw0 = opt.w, aw1 = opt.w;
			a.score = a.truesc = -1;
			a.rid = xj.rid;
			if (s.qbeg) {
				const ssg_xres_t o = xres_l[gid];
				aw0 = o.aw; a.score = o.score;
				if (o.gscore <= 0 || o.gscore <= a.score - opt.pen_clip5) { a.qb = s.qbeg - o.qle; a.rb = s.rbeg - o.tle; a.truesc = a.score; }
				else { a.qb = 0; a.rb = s.rbeg - o.gtle; a.truesc = o.gscore; }
			} else { a.score = a.truesc = s.len * opt.a; a.qb = 0; a.rb = s.rbeg; }
			if (s.qbeg + s.len != l_query) {
				const int qe = s.qbeg + s.len, sc0 = a.score;
				const int re = (int)(s.rbeg + s.len - xj.rmax0);
				const ssg_xres_t o = xres_r[gid];
				aw1 = o.aw; a.score = o.score;
				if (o.gscore <= 0 || o.gscore <= a.score - opt.pen_clip3) { a.qe = qe + o.qle; a.re = xj.rmax0 + re + o.tle; a.truesc += a.score - sc0; }
				else { a.qe = l_query; a.re = xj.rmax0 + re + o.gtle; a.truesc += o.gscore - sc0; }
			} else { a.qe = l_query; a.re = s.rbeg + s.len; }
			a.seedcov = 0;
			for (i = 0; i < cn; ++i) {
				const ssg_seed_t t = cn == 1 ? s : seeds[cs[i]];
				if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) a.seedcov += t.len;
			}
			a.w = aw0 > aw1 ? aw0 : aw1;
			a.seedlen0 = s.len;
			a.frac_rep = xj.frac_rep;
			av[av_n++] = a;
		}
	}
	/* upstream mem_sort_dedup_patch, serial; a pair of regions that would reach mem_patch_reg's alignment sends the read to the wave kernel */
	if (av_n > 1) {
		ssg_introsort(av, (long)av_n, ssg_reg_re_lt());
		for (int i = 0; i < av_n; ++i) av[i].n_comp = 1;
		for (int i = 1; i < av_n; ++i) {
			ssg_alnreg_t *p = &av[i];
			if (p->rid != av[i-1].rid || p->rb >= av[i-1].re + opt.max_chain_gap) continue;
			for (int j = i - 1; j >= 0 && p->rid == av[j].rid && p->rb < av[j].re + opt.max_chain_gap; --j) {
				ssg_alnreg_t *q = &av[j];
				if (q->qe == q->qb) continue;
				const int64_t or_ = q->re - p->rb, oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
				const int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb, mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
				if (or_ > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq) {
					if (p->score < q->score) { p->qe = p->qb; break; }
					else q->qe = q->qb;
				} else if (q->rb < p->rb) {
					ssg_sdp_key_t kq, kp; kq.re = q->re; kq.rb = q->rb; kq.qb = q->qb; kq.qe = q->qe; kp.re = p->re; kp.rb = p->rb; kp.qb = p->qb; kp.qe = p->qe;
					if (ssg_patch_candidate(opt, l_pac, kq, kp)) SSG_C2A_TODO();   /* regs[] is rebuilt from scratch by the wave kernel */
				}
			}
		}
		int m = 0;
		for (int t = 0; t < av_n; ++t) if (av[t].qe > av[t].qb) { if (m != t) av[m++] = av[t]; else ++m; }
		av_n = m;
		ssg_introsort(av, (long)av_n, ssg_reg_sc_lt());
		for (int t = 1; t < av_n; ++t) if (av[t].score == av[t-1].score && av[t].rb == av[t-1].rb && av[t].qb == av[t-1].qb) av[t].qe = av[t].qb;
		int mm = av_n < 1 ? av_n : 1;
		for (int t = 1; t < av_n; ++t) if (av[t].qe > av[t].qb) { if (mm != t) av[mm] = av[t]; ++mm; }
		av_n = mm;
	}   /* upstream returns n <= 1 untouched */
	n_reg[r] = av_n; err[r] = 0;
	if (sdp_fixed) sdp_fixed[r] = 1;
#undef SSG_C2A_TODO
}

/* grid-strided: every resident wavefront owns one LDS window and one SSG_TWIN_GLB slab of tglb */
template <bool WIDE>
__global__ void __launch_bounds__(256, SSG_C2A_WAVES_PER_SIMD) ssg_k_chain2aln(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const uint8_t *seq, const int64_t *read_off,
                                const int64_t *seed_off, const ssg_seed_t *seeds, const ssg_chain_t *chains, const int32_t *order,
                                const int32_t *chain_seeds, const int32_t *n_chain, uint64_t *srt_all, ssg_alnreg_t *regs, int32_t *n_reg,
                                uint8_t *tglb, int32_t *err, unsigned long long *cells, const int32_t *work_order, unsigned int *queue, int tune,
                                ssg_sdp_big_t *sdpbig, ssg_alnreg_t *bcopy,
                                const int64_t *chain_off, const ssg_xjob_t *xjobs, const ssg_xres_t *xres_l, const ssg_xres_t *xres_r, const int32_t *todo_list, const unsigned int *n_todo, uint8_t *sdp_fixed)
{
	__shared__ uint8_t tlds[SSG_WAVES_PER_WG][SSG_TWIN_LDS];
	__shared__ ssg_sdp_small_t sdp[SSG_WAVES_PER_WG];
	__shared__ uint16_t hq_[SSG_WAVES_PER_WG][SSG_SDP_BIG];
	__shared__ uint64_t lk_[SSG_WAVES_PER_WG][3 * SSG_C2A_LKEYS];
	const int wslot = (int)(threadIdx.x >> 6);
	const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + wslot;
	unsigned long long nc = 0, ph[8] = {0,0,0,0,0,0,0,0};
	const unsigned long long k0 = tune ? ssg_clock() : 0;
	for (;;) { /* waves pull reads from a heaviest-first list: the per-read work is heavy-tailed (repeats) */
		const long k = wv_queue_pop(queue);
		if (k >= (todo_list ? (long)*n_todo : (long)n_reads)) break;
		wv_chain2aln_read<WIDE>(ix, opt, todo_list ? todo_list[k] : work_order ? work_order[k] : k, seq, read_off, seed_off, seeds, chains, order, chain_seeds, n_chain, srt_all, regs, n_reg,
		                  tlds[wslot], tglb + wave0 * (long)SSG_TWIN_GLB, err, &nc, SSG_TUNING && tune ? ph : 0, &sdp[wslot], sdpbig + wave0, bcopy + wave0 * (long)SSG_SDP_BIG, chain_off, xjobs, xres_l, xres_r, sdp_fixed, hq_[wslot], lk_[wslot]);
	}
	if (wv_lane() == 0 && cells) atomicAdd(cells, nc);
	if (SSG_TUNING && tune && wv_lane() == 0) { /* tuning: window+seed sort, containment scan, extension, re-sort, wave total; #chains, #extended seeds, #regions */
		ph[4] = ssg_clock() - k0;
		for (int t = 0; t < 8; ++t) atomicAdd(&ssg_dbg_cyc[16 + t], ph[t]);
	}
}
#endif
