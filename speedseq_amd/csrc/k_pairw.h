/*
 * k_pairw.h -- primary marking and pairing for pairs with LONG region lists, one wavefront per pair
 * (same functions as the lane-per-pair path of k_pair.h: upstream
 * mem_mark_primary_se, mem_pair; the decision that follows, ssg_pair_decide, is shared).
 *
 * A pair whose ends fall in a repeat family arrives here with 10^2..10^3 regions per end.  One lane
 * sorting them (twice) and walking them at HBM latency is the whole tail of the pairing kernel, so for
 * these pairs the 64 lanes share the work:
 *   - both sorts have distinct keys -- (score, hash) with hash_64 a bijection of the index, and
 *     (position, score|index|strand|end) -- so any correct sort yields upstream's order: rank sort,
 *     rank = number of smaller keys, O(n^2/64) per lane on 16-byte keys;
 *   - mem_mark_primary's scan (region i against the primaries found so far, stop at the first overlap)
 *     takes 64 regions per step: every lane finds its first overlapping primary, the first lane without
 *     one becomes a primary and the lanes behind it are re-examined; `sub' takes the score of the first
 *     region that hit a primary and `sub_n' a ballot count, as the serial loop would have left them;
 *   - mem_pair's candidate walk runs one lane per region (backwards over the position-sorted list until
 *     the insert-size bound is passed); only the best two (q, hash) keys and a histogram of q are kept,
 *     which is all that upstream reads back from its sorted candidate array (best, second best, and the
 *     number of candidates within one mismatch/gap of the second best).
 */
#ifndef SSG_K_PAIRW_H
#define SSG_K_PAIRW_H
#include "k_pair.h"

#define SSG_PW_NCAP 4096   /* regions per end */
#define SSG_PW_ZCAP 64     /* primaries per end (mutually non-overlapping on the read: a handful) */
#define SSG_PW_QCAP 1024   /* histogram of pair scores */
struct ssg_pw_slab_t { ssg_pair64_t key[2 * SSG_PW_NCAP], srt[2 * SSG_PW_NCAP]; int32_t ord[SSG_PW_NCAP]; ssg_alnreg_t tmp[SSG_PW_NCAP]; ssg_pair64_t u[1024]; };
struct ssg_pw_lds_t { int32_t zqb[SSG_PW_ZCAP], zqe[SSG_PW_ZCAP], zsc[SSG_PW_ZCAP], zidx[SSG_PW_ZCAP], zsub[SSG_PW_ZCAP], zsubn[SSG_PW_ZCAP]; uint32_t hist[SSG_PW_QCAP]; };

SSG_DEVFN bool ssg_p128_less(const ssg_pair64_t &a, const ssg_pair64_t &b) { return (a.x < b.x) | ((a.x == b.x) & (a.y < b.y)); }

/* out[rank of key[i]] = i for distinct keys (below the all-ones key).  A block of 64 keys is loaded once, one per lane, and handed round by v_readlane: no
 * memory in the inner loop (k_sdp.h wv_rank_u64 has the measurement that led here). */
SSG_DEVFN void wv_rank_sort128(const ssg_pair64_t *key, int n, int32_t *out_idx, ssg_pair64_t *out_key)
{
	const int lane = wv_lane();
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int i = i0 + lane;
		ssg_pair64_t me; me.x = me.y = ~0ull;
		if (i < n) me = key[i];
		int r = 0;
		for (int j0 = 0; j0 < n; j0 += 64) {
			ssg_pair64_t kl; kl.x = kl.y = ~0ull;
			if (j0 + lane < n) kl = key[j0 + lane];
			SSG_UNROLL for (int t = 0; t < 64; ++t) {
				const uint64_t kx = (uint64_t)wv_get64((long long)kl.x, t), ky = (uint64_t)wv_get64((long long)kl.y, t);
				r += (kx < me.x) | ((kx == me.x) & (ky < me.y));
			}
		}
		if (i < n) { if (out_idx) out_idx[r] = i; if (out_key) out_key[r] = me; }
	}
}

SSG_DEVFN int wv_mark_primary_se(const ssg_mem_opt_t &opt, int n, ssg_alnreg_t *a, int64_t id, ssg_pw_slab_t *S, ssg_pw_lds_t *L, int *err)
{	/* upstream mem_mark_primary_se + _core (ALT-free), all lanes */
	const int lane = wv_lane();
	if (n == 0) return 0;
	ssg_wave_memsync();
	for (int i = lane; i < n; i += 64) {
		ssg_alnreg_t *r = &a[i];
		const uint64_t h = ssg_hash64((uint64_t)(id + i));
		r->sub = r->alt_sc = 0; r->secondary = r->secondary_all = -1; r->hash = h;
		ssg_pair64_t k; k.x = (uint64_t)((int64_t)2147483647 - r->score); k.y = h;   /* score descending, hash ascending */
		S->key[i] = k;
	}
	ssg_wave_memsync();
	wv_rank_sort128(S->key, n, S->ord, 0);
	ssg_wave_memsync();
	for (int k = lane; k < n; k += 64) S->tmp[k] = a[S->ord[k]];
	ssg_wave_memsync();
	for (int k = lane; k < n; k += 64) a[k] = S->tmp[k];
	ssg_wave_memsync();
	int tmp = opt.a + opt.b;
	tmp = opt.o_del + opt.e_del > tmp ? opt.o_del + opt.e_del : tmp;
	tmp = opt.o_ins + opt.e_ins > tmp ? opt.o_ins + opt.e_ins : tmp;
	int zn = 0;
	for (int i0 = 0; i0 < n; ) {
		const int i = i0 + lane;
		int qb = 0, qe = 0, sc = 0, subn0 = 0;
		if (i < n) { qb = a[i].qb; qe = a[i].qe; sc = a[i].score; subn0 = a[i].sub_n; }
		int hit = -1;
		for (int k = 0; k < zn; ++k) {
			const int jb = L->zqb[k], je = L->zqe[k];
			const int b_max = jb > qb ? jb : qb, e_min = je < qe ? je : qe;
			if (e_min > b_max) {
				const int min_l = qe - qb < je - jb ? qe - qb : je - jb;
				if (hit < 0 && e_min - b_max >= min_l * opt.mask_level) hit = k;
			}
		}
		const unsigned long long none = wv_ballot(i < n && hit < 0);
		const int F = none ? (int)__builtin_ctzll(none) : 64;   /* first region of this step that starts a new primary */
		const int mine = i < n && lane < F;                      /* regions ahead of it are settled by this step */
		for (int k = 0; k < zn; ++k) {
			const unsigned long long m = wv_ballot(mine && hit == k);
			if (!m) continue;
			const int first_sc = wv_get(sc, (int)__builtin_ctzll(m));
			const int cnt = __popcll(wv_ballot(mine && hit == k && L->zsc[k] - sc <= tmp));
			ssg_wave_ldssync();
			if (lane == 0) { if (L->zsub[k] == 0) L->zsub[k] = first_sc; L->zsubn[k] += cnt; }
			ssg_wave_ldssync();
		}
		if (mine) { const int j = L->zidx[hit]; a[i].secondary = j; a[i].secondary_all = j; }
		if (F < 64) {
			if (zn >= SSG_PW_ZCAP) { *err = 5; return n; }
			const int fqb = wv_get(qb, F), fqe = wv_get(qe, F), fsc = wv_get(sc, F), fsn = wv_get(subn0, F);
			ssg_wave_ldssync();
			if (lane == 0) { L->zqb[zn] = fqb; L->zqe[zn] = fqe; L->zsc[zn] = fsc; L->zidx[zn] = i0 + F; L->zsub[zn] = 0; L->zsubn[zn] = fsn; }
			ssg_wave_ldssync();
			++zn;
			i0 += F + 1;
		} else i0 += 64;
	}
	ssg_wave_ldssync();
	if (lane < zn) { ssg_alnreg_t *r = &a[L->zidx[lane]]; r->sub = L->zsub[lane]; r->sub_n = L->zsubn[lane]; }
	ssg_wave_memsync();
	return n;
}

SSG_DEVFN int wv_mem_pair(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const ssg_pestat_t *pes, ssg_alnreg_t *const a[2], int id,
                          int *sub, int *n_sub, int z[2], const int n_pri[2], ssg_pw_slab_t *S, ssg_pw_lds_t *L, int *err)
{	/* upstream mem_pair, all lanes */
	const int lane = wv_lane();
	const int64_t l_pac = ix.l_pac;
	const int N = n_pri[0] + n_pri[1];
	ssg_wave_memsync();
	for (int t = lane; t < N; t += 64) {
		const int r = t >= n_pri[0], i = r ? t - n_pri[0] : t;
		const ssg_alnreg_t &e = a[r][i];
		ssg_pair64_t key;
		key.x = (uint64_t)(e.rb < l_pac ? e.rb : (l_pac << 1) - 1 - e.rb);
		key.x = (uint64_t)e.rid << 32 | (key.x - (uint64_t)ix.ctg_off[e.rid]);
		key.y = (uint64_t)e.score << 32 | (uint64_t)(i << 2) | (uint64_t)((e.rb >= l_pac) << 1) | (uint64_t)r;
		S->key[t] = key;
	}
	for (int t = lane; t < SSG_PW_QCAP; t += 64) L->hist[t] = 0;
	ssg_wave_memsync();
	wv_rank_sort128(S->key, N, 0, S->srt);
	ssg_wave_memsync();
	const ssg_pair64_t *v = S->srt;
	ssg_pair64_t b1, b2; b1.x = b1.y = b2.x = b2.y = 0;   /* best and second-best candidate keys of this lane */
	int nb = 0, qovf = 0;
	for (int i = lane; i < N; i += 64) {
		const ssg_pair64_t vi = v[i];
		for (int r = 0; r < 2; ++r) {
			const int dir = r << 1 | (int)(vi.y >> 1 & 1);
			if (pes[dir].failed) continue;
			const int which = r << 1 | (int)((vi.y & 1) ^ 1);
			for (int k = i - 1; k >= 0; --k) {
				const ssg_pair64_t vk = v[k];
				const int64_t dist = (int64_t)vi.x - (int64_t)vk.x;
				if (dist > pes[dir].high) break;   /* sorted by position: nothing further back can be in range */
				if ((int)(vk.y & 3) != which) continue;
				if (dist < pes[dir].low) continue;
				const double ns = (dist - pes[dir].avg) / pes[dir].std;
				int q = (int)((vi.y >> 32) + (vk.y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt.a + .499);
				if (q < 0) q = 0;
				ssg_pair64_t u;
				u.y = (uint64_t)k << 32 | (uint64_t)i;
				u.x = (uint64_t)q << 32 | (ssg_hash64(u.y ^ (uint64_t)(int64_t)(id << 8)) & 0xffffffffU);
				if (q >= SSG_PW_QCAP) { qovf = 1; q = SSG_PW_QCAP - 1; }
				atomicAdd(&L->hist[q], 1u);
				if (nb == 0 || ssg_p128_less(b1, u)) { b2 = b1; b1 = u; nb = nb < 2 ? nb + 1 : 2; }
				else if (nb == 1 || ssg_p128_less(b2, u)) { b2 = u; nb = 2; }
			}
		}
	}
	if (wv_ballot(qovf)) *err = 6;
	ssg_wave_ldssync();
	/* wave-wide best and second best */
	ssg_pair64_t g1 = b1; int have = nb > 0;
	if (!have) { g1.x = 0; g1.y = 0; }
	for (int d = 1; d < 64; d <<= 1) {
		ssg_pair64_t o; o.x = (uint64_t)wv_shfl64_xor((long long)g1.x, d); o.y = (uint64_t)wv_shfl64_xor((long long)g1.y, d);
		const int oh = wv_shfl(have, wv_lane() ^ d);
		if (oh && (!have || ssg_p128_less(g1, o))) { g1 = o; have = 1; }
	}
	if (!have) { *sub = 0; *n_sub = 0; return 0; }
	/* my candidate for the second place: my best unless it is the winner, then my second */
	const int i_won = nb > 0 && b1.x == g1.x && b1.y == g1.y;
	ssg_pair64_t g2 = i_won ? b2 : b1; int have2 = i_won ? nb > 1 : nb > 0;
	if (!have2) { g2.x = 0; g2.y = 0; }
	for (int d = 1; d < 64; d <<= 1) {
		ssg_pair64_t o; o.x = (uint64_t)wv_shfl64_xor((long long)g2.x, d); o.y = (uint64_t)wv_shfl64_xor((long long)g2.y, d);
		const int oh = wv_shfl(have2, wv_lane() ^ d);
		if (oh && (!have2 || ssg_p128_less(g2, o))) { g2 = o; have2 = 1; }
	}
	int tmp = opt.a + opt.b;
	tmp = tmp > opt.o_del + opt.e_del ? tmp : opt.o_del + opt.e_del;
	tmp = tmp > opt.o_ins + opt.e_ins ? tmp : opt.o_ins + opt.e_ins;
	const int i = (int)(g1.y >> 32), k = (int)(g1.y << 32 >> 32);
	z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
	z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
	const int ret = (int)(g1.x >> 32);
	*sub = have2 ? (int)(g2.x >> 32) : 0;
	/* number of candidates other than the best whose score is within tmp of the second best */
	int c = 0;
	for (int q = lane; q < SSG_PW_QCAP; q += 64) if (*sub - q <= tmp) c += (int)L->hist[q];
	c = wv_sum(c);
	*n_sub = c - 1;
	return ret;
}

/* one wave per pair, pairs work_order[0 .. n_heavy) from a queue */
__global__ void __launch_bounds__(256) ssg_k_pair_final_wave(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_heavy, int64_t id0,
                                 const int64_t *reg_off, ssg_alnreg_t *regs, const int32_t *n_reg, const int32_t *pair_batch, const ssg_pestat_t *pes_all,
                                 int32_t *zbuf, ssg_pair64_t *vbuf, ssg_pw_slab_t *slabs,
                                 const int64_t *req_off, ssg_alnreq_t *req, int32_t *n_req, int32_t *err, const int32_t *work_order, unsigned int *queue)
{
	__shared__ ssg_pw_lds_t lds[SSG_WAVES_PER_WG];
	const int wslot = (int)(threadIdx.x >> 6);
	ssg_pw_slab_t *S = slabs + ((long)blockIdx.x * (blockDim.x >> 6) + wslot);
	ssg_pw_lds_t *L = &lds[wslot];
	for (;;) {
		const long pq = wv_queue_pop(queue);
		if (pq >= n_heavy) break;
		const long p = work_order[pq];
		const ssg_pestat_t *pes = pes_all + (long)pair_batch[p] * 4;
		const int64_t id = id0 + p;
		ssg_alnreg_t *a[2] = { regs + reg_off[2*p], regs + reg_off[2*p+1] };
		const int an[2] = { n_reg[2*p], n_reg[2*p+1] };
		int32_t *z0 = zbuf + reg_off[2*p];
		ssg_pair64_t *v = vbuf + reg_off[2*p];
		int n_pri[2], z[2] = {0, 0}, o = 0, subo = 0, n_sub = 0, myerr = 0;
		ssg_alnreq_t *rq[2] = { req + req_off[2*p], req + req_off[2*p+1] };
		if (an[0] > SSG_PW_NCAP || an[1] > SSG_PW_NCAP) { /* beyond the slab: the serial routines on one lane */
			ssg_wave_memsync();
			if (wv_lane() == 0) {
				n_pri[0] = ssg_mark_primary_se(opt, an[0], a[0], id << 1 | 0, z0, (int32_t*)v);
				n_pri[1] = ssg_mark_primary_se(opt, an[1], a[1], id << 1 | 1, z0, (int32_t*)v);
				if (n_pri[0] && n_pri[1] && !(opt.flag & SSG_F_NOPAIRING)) o = ssg_mem_pair(ix, opt, pes, a, (int)id, &subo, &n_sub, z, n_pri, v, S->u, 1024, &myerr);
				ssg_pair_decide(ix, opt, pes, p, a, an, n_pri, o, subo, n_sub, z, z0, reg_off, rq, n_req);
				if (myerr) err[p] = myerr;
			}
			ssg_wave_memsync();
			continue;
		}
		n_pri[0] = wv_mark_primary_se(opt, an[0], a[0], id << 1 | 0, S, L, &myerr);
		n_pri[1] = wv_mark_primary_se(opt, an[1], a[1], id << 1 | 1, S, L, &myerr);
		if (n_pri[0] && n_pri[1] && !myerr && !(opt.flag & SSG_F_NOPAIRING)) o = wv_mem_pair(ix, opt, pes, a, (int)id, &subo, &n_sub, z, n_pri, S, L, &myerr);
		ssg_wave_memsync();
		if (wv_lane() == 0) {
			if (!myerr) ssg_pair_decide(ix, opt, pes, p, a, an, n_pri, o, subo, n_sub, z, z0, reg_off, rq, n_req);
			else { n_req[2*p] = n_req[2*p+1] = 0; err[p] = myerr; }
		}
		ssg_wave_memsync();
	}
}
#endif
