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

/* ---- klib's B-tree as upstream mem_chain uses it (kbtree.h: KBTREE_INIT(chn, mem_chain_t, chain_cmp), kb_init(chn, 512): minimum degree t = 5 for the 40-byte
 * mem_chain_t, up to 9 chains a node; [RECALL: kbtree.h is not in the reference tree], restated in oracle/orc_mem.c, which says where it parts from the (pos, sec)
 * order: only for a read with more than 9 chains AND two chains at one position).  Every chaining kernel flags such reads (ssg_kbflag); ssg_k_chain_kb then redoes
 * them on this tree: nodes of 21 words in a slab of the caller -- [0] internal? [1] #keys [2..10] keys (chain ids) [11..20] children (node numbers). ---- */
#define SSG_KB_T 5
#define SSG_KB_NODE 21
struct ssg_kb_t { int32_t *nd; int n_nodes, cap_nodes, root; const ssg_chain_t *ch; int ovf; };
SSG_DEVFN int ssg_kb_new(ssg_kb_t &b, int internal) { if (b.n_nodes >= b.cap_nodes) { b.ovf = 1; return 0; } const int x = b.n_nodes++; int32_t *p = b.nd + (long)x * SSG_KB_NODE; p[0] = internal; p[1] = 0; return x; }
SSG_DEVFN int ssg_kb_getp(const ssg_kb_t &b, int x, int64_t pos, int *r)
{	/* __kb_getp_aux: the first key >= pos of node x, one back when that key is greater; -1 for an empty node */
	const int32_t *p = b.nd + (long)x * SSG_KB_NODE;
	int begin = 0, end = p[1];
	if (p[1] == 0) return -1;
	while (begin < end) { const int mid = (begin + end) >> 1; if (b.ch[p[2 + mid]].pos < pos) begin = mid + 1; else end = mid; }
	if (begin == p[1]) { *r = 1; return p[1] - 1; }
	const int64_t kp = b.ch[p[2 + begin]].pos;
	*r = (pos > kp) - (pos < kp);
	if (*r < 0) --begin;
	return begin;
}
SSG_DEVFN int ssg_kb_lower(const ssg_kb_t &b, int64_t pos)
{	/* kb_intervalp's lower bound: a chain with the largest position <= pos, or -1 */
	int x = b.root, lower = -1;
	for (;;) {
		const int32_t *p = b.nd + (long)x * SSG_KB_NODE;
		int r = 0; const int i = ssg_kb_getp(b, x, pos, &r);
		if (i >= 0 && r == 0) return p[2 + i];
		if (i >= 0) lower = p[2 + i];
		if (!p[0]) return lower;
		x = p[11 + i + 1];
	}
}
SSG_DEVFN void ssg_kb_split(ssg_kb_t &b, int x, int i, int y)
{	/* __kb_split: child y of x (the i-th) is full; its upper t - 1 keys go to a new node, its median into x */
	int32_t *py = b.nd + (long)y * SSG_KB_NODE;
	const int z = ssg_kb_new(b, py[0]);
	int32_t *px = b.nd + (long)x * SSG_KB_NODE, *pz = b.nd + (long)z * SSG_KB_NODE;
	pz[1] = SSG_KB_T - 1;
	for (int k = 0; k < SSG_KB_T - 1; ++k) pz[2 + k] = py[2 + SSG_KB_T + k];
	if (py[0]) for (int k = 0; k < SSG_KB_T; ++k) pz[11 + k] = py[11 + SSG_KB_T + k];
	py[1] = SSG_KB_T - 1;
	for (int k = px[1]; k > i; --k) px[11 + k + 1] = px[11 + k];
	px[11 + i + 1] = z;
	for (int k = px[1] - 1; k >= i; --k) px[2 + k + 1] = px[2 + k];
	px[2 + i] = py[2 + SSG_KB_T - 1];
	++px[1];
}
SSG_DEVFN void ssg_kb_put(ssg_kb_t &b, int k)
{	/* kb_putp */
	const int64_t pos = b.ch[k].pos;
	if (b.nd[(long)b.root * SSG_KB_NODE + 1] == 2 * SSG_KB_T - 1) {
		const int r = b.root, s = ssg_kb_new(b, 1);
		b.nd[(long)s * SSG_KB_NODE + 11] = r; b.root = s;
		ssg_kb_split(b, s, 0, r);
	}
	int x = b.root;
	while (!b.ovf) {
		int32_t *p = b.nd + (long)x * SSG_KB_NODE;
		int r = 0;
		if (!p[0]) {
			const int i = ssg_kb_getp(b, x, pos, &r);
			for (int t = p[1] - 1; t > i; --t) p[2 + t + 1] = p[2 + t];
			p[2 + i + 1] = k; ++p[1];
			return;
		}
		int i = ssg_kb_getp(b, x, pos, &r) + 1;
		if (b.nd[(long)p[11 + i] * SSG_KB_NODE + 1] == 2 * SSG_KB_T - 1) {
			ssg_kb_split(b, x, i, p[11 + i]);
			if (pos > b.ch[p[2 + i]].pos) ++i;
		}
		x = p[11 + i];
	}
}
SSG_DEVFN int ssg_kb_inorder(const ssg_kb_t &b, int32_t *out)
{	/* __kb_traverse */
	int stk_x[16], stk_i[16], top = 0, n = 0;
	stk_x[0] = b.root; stk_i[0] = 0;
	while (top >= 0) {
		const int32_t *p = b.nd + (long)stk_x[top] * SSG_KB_NODE;
		const int i = stk_i[top];
		if (!p[0]) { for (int k = 0; k < p[1]; ++k) out[n++] = p[2 + k]; --top; continue; }
		if (i > p[1]) { --top; continue; }
		if (i > 0) out[n++] = p[2 + i - 1];
		stk_i[top] = i + 1;
		if (top + 1 < 16) { ++top; stk_x[top] = p[11 + i]; stk_i[top] = 0; }
	}
	return n;
}
/* a read whose chains may lie differently in upstream's tree: more than 9 of them and two at one position */
SSG_DEVFN void ssg_kbflag(int32_t *kbflag, long r, int n_chains, int n_dup) { if (kbflag && n_chains > 9 && n_dup > 0) kbflag[r] = 1; }

/*
 * The chains of one read (upstream mem_chain + mem_chain_flt) on the slices the caller hands over (global memory in ssg_k_chain; a form with the lane's
 * part of LDS was built in round 4 and measured slower): sd[] / srid[] the read's ns seeds, iv[] its ni intervals, ch[] / ord[] / kp[] / cs[] work arrays of ns entries.
 * Returns the number of chains that survive the filter; ord[0..n) = their ids in upstream's final order; for each, ch[id].first_seed is
 * rewritten to s0 + (offset into cs[]) where its n seed ids (s0 + index, insertion order) lie.
 */
SSG_DEVFN int ssg_chain_one(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, const int len, const int ns, ssg_seed_t *sd, const int32_t *srid, const int ni, const ssg_intv_t *iv,
                            ssg_chain_t *ch, int32_t *ord, int32_t *kp, int32_t *cs, const long s0, const int dbg_phase,
                            int32_t *kb_slab = 0, int kb_nodes = 0 /* != 0: the chains in klib's B-tree (nodes in kb_slab) instead of the (pos, sec) tree */, int *flag_out = 0, int *err_out = 0)
{
	int nc = 0, root = -1, ins_ctr = 0, i, k;
	ssg_kb_t kb; kb.nd = kb_slab; kb.n_nodes = 0; kb.cap_nodes = kb_nodes; kb.root = 0; kb.ch = ch; kb.ovf = 0;
	if (kb_slab) kb.root = ssg_kb_new(kb, 0);
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
		if (kb_slab) lower = nc ? ssg_kb_lower(kb, rbeg) : -1;
		else while (cur >= 0) { /* floor of (rbeg, SEC_FIRST) */
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
			if (kb_slab) ssg_kb_put(kb, nc);
			else if (root < 0) root = nc;
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
	if (flag_out) *flag_out = nc > 9 && ins_ctr > 0;
	if (err_out && kb.ovf) *err_out = 1;
	if (dbg_phase == 1) return 0;
	if (kb_slab && !kb.ovf) { const int m = ssg_kb_inorder(kb, ord); if (m != nc && err_out) *err_out = 1; }
	else {
		for (i = 0; i < nc; ++i) ord[i] = i;
		ssg_chain_key_lt lt = { ch }; ssg_introsort(ord, (long)nc, lt);   /* == B-tree in-order traversal while the tree is one leaf, or the positions are distinct */
	}
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
                            const int32_t *work_order, int32_t *kbflag)
{
	long r = r_first + (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	if (work_order) r = work_order[r];   /* reads sorted by seed count: the lanes of a wave get similar work */
	const int len = (int)(read_off[r+1] - read_off[r]);
	const long s0 = seed_off[r]; const int ns = (int)(seed_off[r+1] - s0);
	int flag = 0;
	n_chain[r] = ssg_chain_one(ix, opt, len, ns, seeds + s0, seed_rid + s0, n_intv[r] > 0 ? n_intv[r] : 0, intv + r * cap, chains + s0, order + s0, kept + s0, chain_seeds + s0, s0, dbg_phase, 0, 0, &flag);
	if (kbflag && flag) kbflag[r] = 1;
}

/* The flagged reads again, their chains in klib's B-tree: one lane per listed read (they are few: none in a million simulated human pairs, every one of a
 * test's constructed reads); the seeds' chain links are reset first (the first pass left its own).  slab_off[k] .. slab_off[k + 1]: the words of read list[k]'s nodes. */
__global__ void __launch_bounds__(64) ssg_k_chain_kb(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_list, const int32_t *list,
                            const int64_t *read_off, const ssg_intv_t *intv, const int32_t *n_intv, int cap,
                            const int64_t *seed_off, ssg_seed_t *seeds, const int32_t *seed_rid,
                            ssg_chain_t *chains, int32_t *order, int32_t *kept, int32_t *chain_seeds, int32_t *n_chain,
                            int32_t *slab, const int64_t *slab_off, int32_t *err)
{
	const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_list) return;
	const long r = list[k];
	const int len = (int)(read_off[r+1] - read_off[r]);
	const long s0 = seed_off[r]; const int ns = (int)(seed_off[r+1] - s0);
	for (int i = 0; i < ns; ++i) seeds[s0 + i].next = -1;
	int e = 0;
	n_chain[r] = ssg_chain_one(ix, opt, len, ns, seeds + s0, seed_rid + s0, n_intv[r] > 0 ? n_intv[r] : 0, intv + r * cap, chains + s0, order + s0, kept + s0, chain_seeds + s0, s0, 0,
	                           slab + slab_off[k], (int)((slab_off[k + 1] - slab_off[k]) / SSG_KB_NODE), 0, &e);
	if (e) atomicMax(err, 1);
}
/* the flagged reads as a list, and the node words each needs (a tree of nc <= ns chains has at most nc / 4 leaves and a third as many inner nodes) */
__global__ void ssg_k_chain_kb_list(int n_reads, const int32_t *kbflag, const int64_t *seed_off, int32_t *list, int32_t *need, unsigned int *n_list)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads || !kbflag[r]) return;
	const unsigned int k = atomicAdd(n_list, 1u);
	list[k] = (int32_t)r;
	need[k] = (int32_t)(((seed_off[r + 1] - seed_off[r]) / 3 + 12) * SSG_KB_NODE);
}

/*
 * The light reads (fewer than 64 seeds: 97 % of the reads, half of the seeds) with the read's whole state in the lane's part of LDS.
 * ssg_k_chain above runs at the rate HBM delivers random 64-byte lines (profiles/r05_pmc_traffic.json: 79 GB fetched per launch for
 * ~3 GB of seeds, chains and lists): every lane walks its own few hundred bytes of chain and seed records, the resident lanes' working
 * sets together exceed L2 many times over, and each touch becomes a line from HBM.  Here a lane reads its seeds ONCE (the algorithmic
 * traffic), chains and filters on LDS, and writes out what the next stage reads: the surviving chains, their order and their seed lists.
 *
 * Layout: 32-bit words, word W of lane l at lds[W * LN + l] (LN lanes a workgroup; lanes at the same W hit different banks); 5.75 words per seed slot:
 *   RLO / RHI [seed]   rbeg of the seed        | after the weights: S[i] = w << 8 | chain (position order, then sorted by weight) / K[k] = kept chain's query begin | end << 9 | w << 18
 *   META [seed]        qbeg | len << 9 | contig << 18 (0x3fff: no contig, the seed is skipped; the host sends indexes with more contigs to ssg_k_chain)
 *   CA [chain]         last seed | left << 8 | right << 16 | #seeds << 24     (a chain's id IS its first seed's index: its position and contig are that seed's)
 *   CB [chain]         w | sec << 16 (0: the first chain at its position; later ones 64 - k, k = 1, 2, ...: newest first, as the (pos, sec) order of ssg_chain_one)
 *                      | after the weights, as bytes: state of sorted chain i / first chain shadowed by kept chain k
 *   NEXT [seed], SUCC [chain] (bytes)   next seed of the chain; next chain in (pos, sec) order
 * Against ssg_chain_one: one descent of the tree per seed instead of two (the place where the floor search falls off the tree is the place
 * where the new chain hangs: both descents compare the same keys the same way); the in-order listing is a linked list kept while
 * inserting (a new chain is the immediate successor of its floor) instead of a sort afterwards; the weight sort replays upstream's
 * introsort on the packed words.  Same chains, same order (tests: against the oracle, and against ssg_k_chain on the same reads).
 */
#define SSG_CL_NONE 0xffu
template <int CAP> struct ssg_cl_cfg {
	static constexpr int W_RLO = 0, W_RHI = CAP, W_META = 2 * CAP, W_CA = 3 * CAP, W_CB = 4 * CAP, W_NEXT = 5 * CAP, W_SUCC = W_NEXT + CAP / 4, W_ORD = W_SUCC + CAP / 4, WORDS = W_ORD + CAP / 4;
};
template <int LN> struct ssg_cl_words_t {   /* a[i] = word (base + i) of this lane; LN lanes share the block */
	uint32_t *w;
	SSG_DEVMEM uint32_t get(int i) const { return w[i * LN]; }
	SSG_DEVMEM void set(int i, uint32_t x) const { w[i * LN] = x; }
};
struct ssg_cl_w_gt { SSG_DEVMEM bool operator()(uint32_t a, uint32_t b) const { return (a >> 8) > (b >> 8); } };
#define SSG_CL_B(word0, e) (((uint8_t*)(lw + ((word0) + ((e) >> 2)) * LN))[(e) & 3])

template <int CAP, int LN /* reads (lanes) per workgroup: 64, or 32 where a whole wave's state would leave a CU's LDS to one workgroup */>
__global__ void __launch_bounds__(64) ssg_k_chain_lds(ssg_index_view_t ix, ssg_mem_opt_t opt, int r_first, int r_end,
                            const int64_t *read_off, const ssg_intv_t *intv, const int32_t *n_intv, int cap,
                            const int64_t *seed_off, const ssg_seed_t *seeds, const int32_t *seed_rid,
                            ssg_chain_t *chains, int32_t *order, int32_t *chain_seeds, int32_t *n_chain, const int32_t *work_order, int32_t *kbflag)
{
	typedef ssg_cl_cfg<CAP> C;
	static_assert(CAP <= 64 && CAP % 4 == 0, "chain ids are 6 bits, byte arrays fill whole words");
	__shared__ uint32_t lds[C::WORDS * LN];
	long r = r_first + (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= r_end) return;
	if (work_order) r = work_order[r];
	uint32_t *lw = lds + threadIdx.x;
	const int len = (int)(read_off[r+1] - read_off[r]);
	const long s0 = seed_off[r]; const int ns = (int)(seed_off[r+1] - s0);
	const int64_t l_pac = ix.l_pac;
	int i, k;
	/* frac_rep (upstream mem_chain head) */
	float frac_rep;
	{
		const int ni = n_intv[r] > 0 ? n_intv[r] : 0;
		const ssg_intv_t *iv = intv + r * cap;
		int b = 0, e = 0, l_rep = 0;
		for (i = 0; i < ni; ++i) {
			const int sb = (int)(iv[i].info >> 32), se = (int)(uint32_t)iv[i].info;
			if (iv[i].x2 <= (uint64_t)opt.max_occ) continue;
			if (sb > e) l_rep += e - b, b = sb, e = se;
			else e = e > se ? e : se;
		}
		l_rep += e - b;
		frac_rep = (float)l_rep / len;
	}
	if (ns <= 0 || ns > CAP) { n_chain[r] = 0; return; }   /* (more than CAP: never sent here) */
	/* the seeds, once: four in flight */
	for (i = 0; i < ns; i += 4) {
		int64_t rb[4]; int q[4], l[4], rd[4];
		SSG_UNROLL for (k = 0; k < 4; ++k) { const int e = i + k < ns ? i + k : ns - 1; const ssg_seed_t *p = seeds + s0 + e; rb[k] = p->rbeg; q[k] = p->qbeg; l[k] = p->len; rd[k] = seed_rid[s0 + e]; }
		SSG_UNROLL for (k = 0; k < 4; ++k) if (i + k < ns) {
			lw[(C::W_RLO + i + k) * LN] = (uint32_t)rb[k]; lw[(C::W_RHI + i + k) * LN] = (uint32_t)((uint64_t)rb[k] >> 32);
			lw[(C::W_META + i + k) * LN] = (uint32_t)q[k] | (uint32_t)l[k] << 9 | (rd[k] < 0 ? 0x3fffu : (uint32_t)rd[k]) << 18;
		}
	}
	/* greedy chaining in seed-visiting order */
	unsigned root = SSG_CL_NONE, head = SSG_CL_NONE; int ins_ctr = 0, n_made = 0;
	for (i = 0; i < ns; ++i) {
		const uint32_t mi = lw[(C::W_META + i) * LN];
		const uint32_t rid_i = mi >> 18;
		if (rid_i == 0x3fffu) continue;
		const int64_t rbeg = (int64_t)((uint64_t)lw[(C::W_RHI + i) * LN] << 32 | lw[(C::W_RLO + i) * LN]);
		const int qbeg = (int)(mi & 511), slen = (int)(mi >> 9 & 511);
		unsigned cur = root, lower = SSG_CL_NONE, par = SSG_CL_NONE; int par_right = 0;
		int64_t lower_pos = 0; uint32_t lower_ca = 0;
		while (cur != SSG_CL_NONE) { /* floor of (rbeg, first) */
			const int64_t pos = (int64_t)((uint64_t)lw[(C::W_RHI + cur) * LN] << 32 | lw[(C::W_RLO + cur) * LN]);
			const uint32_t ca = lw[(C::W_CA + cur) * LN], cb = lw[(C::W_CB + cur) * LN];
			const bool go_right = pos < rbeg || (pos == rbeg && (cb >> 16) == 0);
			par = cur; par_right = go_right;
			if (go_right) { lower = cur; lower_pos = pos; lower_ca = ca; cur = ca >> 16 & 255; } else cur = ca >> 8 & 255;
		}
		bool merged = false;
		if (lower != SSG_CL_NONE) { /* upstream test_and_merge against the floor chain */
			const uint32_t mf = lw[(C::W_META + lower) * LN];
			const unsigned ls = lower_ca & 255;
			const uint32_t ml = lw[(C::W_META + ls) * LN];
			const int64_t l_rbeg = (int64_t)((uint64_t)lw[(C::W_RHI + ls) * LN] << 32 | lw[(C::W_RLO + ls) * LN]);
			const int f_q = (int)(mf & 511), l_q = (int)(ml & 511), l_len = (int)(ml >> 9 & 511);
			if (rid_i != mf >> 18) merged = false;
			else if (qbeg >= f_q && qbeg + slen <= l_q + l_len && rbeg >= lower_pos && rbeg + slen <= l_rbeg + l_len) merged = true;
			else if ((l_rbeg < l_pac || lower_pos < l_pac) && rbeg >= l_pac) merged = false;
			else {
				const int64_t x = qbeg - l_q, y = rbeg - l_rbeg;
				if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - l_len < opt.max_chain_gap && y - l_len < opt.max_chain_gap) {
					SSG_CL_B(C::W_NEXT, ls) = (uint8_t)i;
					lw[(C::W_CA + lower) * LN] = ((lower_ca & ~255u) | (uint32_t)i) + (1u << 24);
					merged = true;
				}
			}
		}
		if (!merged) { /* a new chain, named after its seed */
			uint32_t sec = 0;
			++n_made;
			if (lower != SSG_CL_NONE && lower_pos == rbeg) { ++ins_ctr; sec = (uint32_t)(64 - ins_ctr); }
			lw[(C::W_CA + i) * LN] = (uint32_t)i | SSG_CL_NONE << 8 | SSG_CL_NONE << 16 | 1u << 24;
			lw[(C::W_CB + i) * LN] = sec << 16;
			if (par == SSG_CL_NONE) root = (unsigned)i;
			else ((uint8_t*)(lw + (C::W_CA + par) * LN))[par_right ? 2 : 1] = (uint8_t)i;
			if (lower == SSG_CL_NONE) { SSG_CL_B(C::W_SUCC, i) = (uint8_t)head; head = (unsigned)i; }
			else { SSG_CL_B(C::W_SUCC, i) = SSG_CL_B(C::W_SUCC, lower); SSG_CL_B(C::W_SUCC, lower) = (uint8_t)i; }
		}
	}
	ssg_kbflag(kbflag, r, n_made, ins_ctr);
	/* upstream mem_chain_weight, both passes in one walk */
	for (unsigned c = head; c != SSG_CL_NONE; c = SSG_CL_B(C::W_SUCC, c)) {
		const int n = (int)(lw[(C::W_CA + c) * LN] >> 24);
		int w1 = 0, w2 = 0, end1 = 0; int64_t end2 = 0; unsigned sid = c;
		for (int j = 0; j < n; ++j) {
			const uint32_t m = lw[(C::W_META + sid) * LN];
			const int64_t srb = (int64_t)((uint64_t)lw[(C::W_RHI + sid) * LN] << 32 | lw[(C::W_RLO + sid) * LN]);
			const int sq = (int)(m & 511), sl = (int)(m >> 9 & 511);
			sid = SSG_CL_B(C::W_NEXT, sid);
			if (sq >= end1) w1 += sl; else if (sq + sl > end1) w1 += sq + sl - end1;
			end1 = end1 > sq + sl ? end1 : sq + sl;
			if (srb >= end2) w2 += sl; else if (srb + sl > end2) w2 += (int)(srb + sl - end2);
			end2 = end2 > srb + sl ? end2 : srb + sl;
		}
		const int w = w2 < w1 ? w2 : w1;
		lw[(C::W_CB + c) * LN] |= (uint32_t)w;   /* a weight never exceeds the read length */
	}
	/* chains in position order with w >= min_chain_weight -> S[] */
	int n_chn = 0;
	for (unsigned c = head; c != SSG_CL_NONE; c = SSG_CL_B(C::W_SUCC, c)) {
		const int w = (int)(lw[(C::W_CB + c) * LN] & 0xffff);
		if (w >= opt.min_chain_weight) { lw[(C::W_RLO + n_chn) * LN] = (uint32_t)w << 8 | c; ++n_chn; }
	}
	int n_out = 0;
	if (n_chn > 0) { /* upstream mem_chain_flt */
		const ssg_cl_words_t<LN> S = { lw + C::W_RLO * LN };
		ssg_introsort_ix(S, n_chn, ssg_cl_w_gt());
		/* byte arrays in the words of CB (its weights live in S now): state of sorted chain i; the first chain kept chain k shadows */
		constexpr int W_ST = C::W_CB, W_KF = C::W_CB + CAP / 4;
		int nk = 0;
		{	/* The chains that cannot `break' against the heaviest one by the weights alone break against none (w descends; both tests are monotone in w): a prefix [0, m) of
			 * the list, all kept.  For these the quadratic loop below leaves behind: `first' of kept chain e = the first later chain that overlaps it; large_ovlp of chain e =
			 * some earlier chain overlaps it -- two scans that stop at their first hit (in a repeat family everything overlaps everything). */
			const int w0 = (int)(lw[(C::W_RLO + 0) * LN] >> 8);
			int m = 0;
			for (; m < n_chn; ++m) {
				const uint32_t me = lw[(C::W_RLO + m) * LN];
				const int wc = (int)(me >> 8);
				if ((wc < w0 * opt.drop_ratio) & (w0 - wc >= opt.min_seed_len << 1)) break;
				const unsigned id = me & 255;
				const unsigned ls = lw[(C::W_CA + id) * LN] & 255;
				const uint32_t ml = lw[(C::W_META + ls) * LN];
				lw[(C::W_RHI + m) * LN] = (lw[(C::W_META + id) * LN] & 511) | ((ml & 511) + (ml >> 9 & 511)) << 9 | (uint32_t)wc << 18;
			}
			for (int e = 0; e < m; ++e) {
				const uint32_t pe = lw[(C::W_RHI + e) * LN];
				const int ib = (int)(pe & 511), ie = (int)(pe >> 9 & 511);
				int f = 0x7f, lo = 0;
				for (int q = e + 1; q < m; ++q) {
					const uint32_t pk = lw[(C::W_RHI + q) * LN];
					const int jb = (int)(pk & 511), je = (int)(pk >> 9 & 511);
					const int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
					if (e_min > b_max) { const int li = ie - ib, lj = je - jb, min_l = li < lj ? li : lj; if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) { f = q; break; } }
				}
				for (int q = e - 1; q >= 0; --q) {
					const uint32_t pk = lw[(C::W_RHI + q) * LN];
					const int jb = (int)(pk & 511), je = (int)(pk >> 9 & 511);
					const int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
					if (e_min > b_max) { const int li = ie - ib, lj = je - jb, min_l = li < lj ? li : lj; if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) { lo = 1; break; } }
				}
				SSG_CL_B(W_KF, e) = (uint8_t)f; SSG_CL_B(W_ST, e) = (uint8_t)(lo ? 2 : 3);
			}
			nk = m;
		}
		for (i = nk; i < n_chn; ++i) {
			const uint32_t me = lw[(C::W_RLO + i) * LN];
			const unsigned id = me & 255; const int wi = (int)(me >> 8);
			const unsigned ls = lw[(C::W_CA + id) * LN] & 255;
			const uint32_t ml = lw[(C::W_META + ls) * LN];
			const int ib = (int)(lw[(C::W_META + id) * LN] & 511), ie = (int)(ml & 511) + (int)(ml >> 9 & 511);
			int large_ovlp = 0;
			for (k = 0; k < nk; ++k) {
				const uint32_t pk = lw[(C::W_RHI + k) * LN];
				const int jb = (int)(pk & 511), je = (int)(pk >> 9 & 511), wj = (int)(pk >> 18);
				const int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
				if (e_min > b_max) {
					const int li = ie - ib, lj = je - jb, min_l = li < lj ? li : lj;
					if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) {
						large_ovlp = 1;
						if (SSG_CL_B(W_KF, k) == 0x7f) SSG_CL_B(W_KF, k) = (uint8_t)i;
						if (wi < wj * opt.drop_ratio && wj - wi >= opt.min_seed_len << 1) break;
					}
				}
			}
			if (k == nk) {
				lw[(C::W_RHI + nk) * LN] = (uint32_t)ib | (uint32_t)ie << 9 | (uint32_t)wi << 18;
				SSG_CL_B(W_KF, nk) = 0x7f; ++nk;
				SSG_CL_B(W_ST, i) = (uint8_t)(i == 0 ? 3 : large_ovlp ? 2 : 3);
			} else SSG_CL_B(W_ST, i) = 0;
		}
		for (k = 0; k < nk; ++k) { const unsigned f = SSG_CL_B(W_KF, k); if (f != 0x7f) SSG_CL_B(W_ST, f) = 1; }
		for (i = k = 0; i < n_chn; ++i) {
			const int kk = SSG_CL_B(W_ST, i);
			if (kk == 0 || kk == 3) continue;
			if (++k >= opt.max_chain_extend) break;
		}
		for (; i < n_chn; ++i) if (SSG_CL_B(W_ST, i) < 3) SSG_CL_B(W_ST, i) = 0;
		/* survivors in weight order: records, order, seed lists */
		int pos = 0;
		for (i = 0; i < n_chn; ++i) {
			const int kk = SSG_CL_B(W_ST, i);
			if (kk == 0) continue;
			const uint32_t me = lw[(C::W_RLO + i) * LN];
			const unsigned id = me & 255;
			const int n = (int)(lw[(C::W_CA + id) * LN] >> 24);
			ssg_chain_t c;
			c.pos = seeds[s0 + id].rbeg; c.first_seed = (int)(s0 + pos); c.last_seed = -1; c.n = n; c.rid = (int)(lw[(C::W_META + id) * LN] >> 18);
			c.w = (int)(me >> 8); c.kept = kk; c.first = -1; c.left = c.right = -1; c.frac_rep = frac_rep; c._pad = 0; c.sec = 0;
			chains[s0 + id] = c;
			order[s0 + n_out++] = (int)id;
			unsigned sid = id;
			for (int j = 0; j < n; ++j) { chain_seeds[s0 + pos++] = (int)(s0 + sid); sid = SSG_CL_B(C::W_NEXT, sid); }
		}
	}
	n_chain[r] = n_out;
}

#endif
