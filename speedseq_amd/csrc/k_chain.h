/*
 * k_chain.h -- gfx950 kernel for seed chaining and chain filtering:
 * upstream mem_chain (test_and_merge over a position-ordered B-tree), mem_chain_weight and
 * mem_chain_flt.  One lane per read; the work is short, branchy and order dependent (greedy
 * insertion in seed-visiting order), so it is kept scalar per read and parallel across the
 * >10^6 reads of a batch.
 *
 * The klib B-tree is replaced by an unbalanced BST keyed by (pos, sec): `sec` reproduces the
 * B-tree's single-leaf order among chains with equal pos (first inserted first, then newest to
 * oldest), so lookups (floor of (rbeg,-inf)) and the final in-order listing are identical for all
 * inputs without duplicate positions and for reads with <= 9 chains.
 */
#ifndef SSG_K_CHAIN_H
#define SSG_K_CHAIN_H
#include "ssg_dev.h"

#define SSG_SEC_FIRST (-2147483647 - 1)

SSG_DEVFN int ssg_test_and_merge(const ssg_mem_opt_t &opt, int64_t l_pac, ssg_chain_t &c, ssg_seed_t *seeds, int sid, int seed_rid)
{	/* upstream test_and_merge; seeds of a chain are linked through ssg_seed_t.next */
	const ssg_seed_t p = seeds[sid];
	const ssg_seed_t last = seeds[c.last_seed], first = seeds[c.first_seed];
	int64_t qend = last.qbeg + last.len, rend = last.rbeg + last.len, x, y;
	if (seed_rid != c.rid) return 0;
	if (p.qbeg >= first.qbeg && p.qbeg + p.len <= qend && p.rbeg >= first.rbeg && p.rbeg + p.len <= rend) return 1;
	if ((last.rbeg < l_pac || first.rbeg < l_pac) && p.rbeg >= l_pac) return 0;
	x = p.qbeg - last.qbeg;
	y = p.rbeg - last.rbeg;
	if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - last.len < opt.max_chain_gap && y - last.len < opt.max_chain_gap) {
		seeds[c.last_seed].next = sid;
		c.last_seed = sid; ++c.n;
		return 1;
	}
	return 0;
}

SSG_DEVFN int ssg_chain_weight(const ssg_chain_t &c, const ssg_seed_t *seeds)
{	/* upstream mem_chain_weight */
	int64_t end; int j, w = 0, tmp, sid;
	for (j = 0, end = 0, sid = c.first_seed; j < c.n; ++j, sid = seeds[sid].next) {
		const ssg_seed_t s = seeds[sid];
		if (s.qbeg >= end) w += s.len;
		else if (s.qbeg + s.len > end) w += (int)(s.qbeg + s.len - end);
		end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
	}
	tmp = w; w = 0;
	for (j = 0, end = 0, sid = c.first_seed; j < c.n; ++j, sid = seeds[sid].next) {
		const ssg_seed_t s = seeds[sid];
		if (s.rbeg >= end) w += s.len;
		else if (s.rbeg + s.len > end) w += (int)(s.rbeg + s.len - end);
		end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
	}
	w = w < tmp ? w : tmp;
	return w < 1<<30 ? w : (1<<30) - 1;
}

struct ssg_chain_key_lt {
	const ssg_chain_t *c;
	/* branch-free on purpose: hipcc -O3 (ROCm 7.2, gfx950) mis-compiles the short-circuit form when it is
	 * inlined into ssg_introsort's partition loops (the wave never leaves the loop; tools/dbg/sorthang.cpp) */
	SSG_DEVMEM bool operator()(int a, int b) const { int64_t pa = c[a].pos, pb = c[b].pos; int sa = c[a].sec, sb = c[b].sec; return (pa < pb) | ((pa == pb) & (sa < sb)); }
};
struct ssg_chain_w_lt {
	const ssg_chain_t *c;
	SSG_DEVMEM bool operator()(int a, int b) const { return c[a].w > c[b].w; }
};

/*
 * The chains of one read (upstream mem_chain + mem_chain_flt) on the slices the caller hands over (global memory in ssg_k_chain; a form with the lane's
 * part of LDS was built in round 4 and measured slower): sd[] / srid[] the read's ns seeds, iv[] its ni intervals, ch[] / ord[] / kp[] / cs[] work arrays of ns entries.
 * Returns the number of chains that survive the filter; ord[0..n) = their ids in upstream's final order; for each, ch[id].first_seed is
 * rewritten to s0 + (offset into cs[]) where its n seed ids (s0 + index, insertion order) lie.
 */
SSG_DEVFN int ssg_chain_one(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const int len, const int ns, ssg_seed_t *sd, const int32_t *srid, const int ni, const ssg_intv_t *iv,
                            ssg_chain_t *ch, int32_t *ord, int32_t *kp, int32_t *cs, const long s0, const int dbg_phase)
{
	int nc = 0, root = -1, ins_ctr = 0, i, k;
	/* frac_rep (upstream mem_chain head) */
	int b = 0, e = 0, l_rep = 0;
	for (i = 0; i < ni; ++i) {
		int sb = (int)(iv[i].info >> 32), se = (int)(uint32_t)iv[i].info;
		if (iv[i].x2 <= (uint64_t)opt.max_occ) continue;
		if (sb > e) l_rep += e - b, b = sb, e = se;
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	/* greedy chaining in seed-visiting order */
	for (i = 0; i < ns; ++i) {
		if (srid[i] < 0) continue;
		int64_t rbeg = sd[i].rbeg;
		int cur = root, lower = -1, to_add = 0;
		while (cur >= 0) { /* floor of (rbeg, SEC_FIRST) */
			if (ch[cur].pos < rbeg || (ch[cur].pos == rbeg && ch[cur].sec == SSG_SEC_FIRST)) { lower = cur; cur = ch[cur].right; }
			else cur = ch[cur].left;
		}
		if (nc) { if (lower < 0 || !ssg_test_and_merge(opt, ix.l_pac, ch[lower], sd, i, srid[i])) to_add = 1; }
		else to_add = 1;
		if (to_add) {
			ssg_chain_t c;
			c.pos = rbeg; c.first_seed = c.last_seed = i; c.n = 1; c.rid = srid[i];
			c.w = 0; c.kept = 0; c.first = -1; c.left = c.right = -1; c.frac_rep = 0; c._pad = 0;
			c.sec = (lower >= 0 && ch[lower].pos == rbeg) ? -(++ins_ctr) : SSG_SEC_FIRST;
			ch[nc] = c;
			if (root < 0) root = nc;
			else {
				cur = root;
				for (;;) {
					bool lt = c.pos < ch[cur].pos || (c.pos == ch[cur].pos && c.sec < ch[cur].sec);
					int *nx = lt ? &ch[cur].left : &ch[cur].right;
					if (*nx < 0) { *nx = nc; break; }
					cur = *nx;
				}
			}
			++nc;
		}
	}
	if (dbg_phase == 1) return 0;
	for (i = 0; i < nc; ++i) ord[i] = i;
	{ ssg_chain_key_lt lt = { ch }; ssg_introsort(ord, (long)nc, lt); } /* == B-tree in-order traversal (keys are unique) */
	if (dbg_phase == 2) return 0;
	float frac_rep = (float)l_rep / len;
	/* upstream mem_chain_flt */
	int n_chn = 0;
	for (i = 0; i < nc; ++i) {
		ssg_chain_t &c = ch[ord[i]];
		c.first = -1; c.kept = 0; c.frac_rep = frac_rep;
		c.w = ssg_chain_weight(c, sd);
		if (c.w >= opt.min_chain_weight) ord[n_chn++] = ord[i];
	}
	if (dbg_phase == 3) return 0;
	int n_out = 0;
	if (n_chn > 0) {
		{ ssg_chain_w_lt lt = { ch }; ssg_introsort(ord, (long)n_chn, lt); }
		/* The kept chains' query interval and weight travel in cs[] (free until the seed lists are flattened), one packed word per kept chain
		 * next to kp[]: the quadratic loop below reads 4 contiguous bytes per kept chain instead of its 56-byte record and two seed
		 * records (three to four cache lines apiece; that loop's re-reads were 17-40x the kernel's algorithmic bytes, profiles/r04_pmc_traffic.json). */
		const bool compact = len < 512;   /* 9-bit query coordinates, weights below 2^14 (a weight never exceeds the read length) */
		int nk = 0;
		ch[ord[0]].kept = 3; kp[nk++] = 0;
		if (compact) { const ssg_chain_t &c0 = ch[ord[0]]; cs[0] = sd[c0.first_seed].qbeg | (sd[c0.last_seed].qbeg + sd[c0.last_seed].len) << 9 | c0.w << 18; }
		for (i = 1; i < n_chn; ++i) {
			int large_ovlp = 0;
			const ssg_chain_t &ci = ch[ord[i]];
			int ib = sd[ci.first_seed].qbeg, ie = sd[ci.last_seed].qbeg + sd[ci.last_seed].len;
			const int wi = ci.w;
			for (k = 0; k < nk; ++k) {
				int jb, je, wj;
				if (compact) { const int pk = cs[k]; jb = pk & 511; je = pk >> 9 & 511; wj = pk >> 18; }
				else { const ssg_chain_t &cj = ch[ord[kp[k]]]; jb = sd[cj.first_seed].qbeg; je = sd[cj.last_seed].qbeg + sd[cj.last_seed].len; wj = cj.w; }
				int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
				if (e_min > b_max) {
					int li = ie - ib, lj = je - jb, min_l = li < lj ? li : lj;
					if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) {
						large_ovlp = 1;
						ssg_chain_t &cj = ch[ord[kp[k]]];
						if (cj.first < 0) cj.first = i;
						if (wi < wj * opt.drop_ratio && wj - wi >= opt.min_seed_len << 1) break;
					}
				}
			}
			if (k == nk) { if (compact) cs[nk] = ib | ie << 9 | wi << 18; kp[nk++] = i; ch[ord[i]].kept = large_ovlp ? 2 : 3; }
		}
		for (i = 0; i < nk; ++i) { const ssg_chain_t &c = ch[ord[kp[i]]]; if (c.first >= 0) ch[ord[c.first]].kept = 1; }
		for (i = k = 0; i < n_chn; ++i) {
			int kk = ch[ord[i]].kept;
			if (kk == 0 || kk == 3) continue;
			if (++k >= opt.max_chain_extend) break;
		}
		for (; i < n_chn; ++i) if (ch[ord[i]].kept < 3) ch[ord[i]].kept = 0;
		for (i = 0; i < n_chn; ++i) if (ch[ord[i]].kept != 0) ord[n_out++] = ord[i];
	}
	if (dbg_phase == 4) return 0;
	/* flatten the seed lists of the surviving chains */
	int pos = 0;
	for (i = 0; i < n_out; ++i) {
		ssg_chain_t &c = ch[ord[i]];
		int sid = c.first_seed, start = pos;
		for (k = 0; k < c.n; ++k, sid = sd[sid].next) cs[pos++] = (int)(s0 + sid);
		c.first_seed = (int)(s0 + start);
	}
	return n_out;
}

/*
 * One lane per read.  Per-read slices (all indexed from seed_off[r], capacity = #seeds of r):
 *   chains[]   chain records, order[] / kept[] int work arrays, chain_seeds[] seed ids per chain.
 * Output: n_chain[r] = #chains surviving the filter; order[0..n) = their ids in upstream's final
 * order; for each, chains[id].first_seed is rewritten to an offset into chain_seeds[] (absolute
 * index) holding its n seed ids (absolute) in insertion order.
 */
__global__ void __launch_bounds__(64) ssg_k_chain(ssg_index_view_t ix, ssg_mem_opt_t opt, int r_first, int n_reads,
                            const int64_t *read_off, const ssg_intv_t *intv, const int32_t *n_intv, int cap,
                            const int64_t *seed_off, ssg_seed_t *seeds, const int32_t *seed_rid,
                            ssg_chain_t *chains, int32_t *order, int32_t *kept, int32_t *chain_seeds, int32_t *n_chain, int dbg_phase,
                            const int32_t *work_order)
{
	long r = r_first + (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	if (work_order) r = work_order[r];   /* reads sorted by seed count: the lanes of a wave get similar work */
	const int len = (int)(read_off[r+1] - read_off[r]);
	const long s0 = seed_off[r]; const int ns = (int)(seed_off[r+1] - s0);
	n_chain[r] = ssg_chain_one(ix, opt, len, ns, seeds + s0, seed_rid + s0, n_intv[r] > 0 ? n_intv[r] : 0, intv + r * cap, chains + s0, order + s0, kept + s0, chain_seeds + s0, s0, dbg_phase);
}

#endif
