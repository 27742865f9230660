/*
 * k_aln.h -- gfx950 kernel turning alignment requests into SAM-ready records:
 * upstream mem_reg2aln -> bwa_gen_cigar2 -> ksw_global2 (+ NM / MD).  One wavefront per request
 * (grid-strided): the banded global DP rows run lane-parallel (k_sw.h), the backtrace and the
 * NM/MD walk are short serial epilogues on lane 0; the direction matrix lives in a per-wave slab.
 */
#ifndef SSG_K_ALN_H
#define SSG_K_ALN_H
#include "k_pair.h"

#define SSG_Z_CAP (192 * 1024)   /* backtrack bytes per resident wave */
#define SSG_ALN_QLDS 320          /* query bytes staged in LDS per wave */

SSG_DEVFN int ssg_infer_bw(int l1, int l2, int score, int a, int q, int r)
{	/* upstream infer_bw */
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	if (w < iabs(l1 - l2)) w = iabs(l1 - l2);
	return w;
}

SSG_DEVFN int ssg_put_int(char *s, int l, int cap, int v, bool wr = true)
{
	char b[12]; int n = 0;
	if (v == 0) b[n++] = '0';
	while (v > 0) { b[n++] = (char)('0' + v % 10); v /= 10; }
	while (n > 0) { if (wr && l < cap) s[l] = b[n-1]; ++l; --n; }
	return l;
}

/* upstream bwa_gen_cigar2 for one region; fills out->cigar/n_cigar/NM/md; returns the score */
template <bool WIDE>
SSG_DEVFN int wv_gen_cigar(const ssg_index_view_t &ix, const ssg_mem_opt_t &opt, int w_, int l_query, const uint8_t *query, int64_t rb, int64_t re,
                           uint8_t *tbuf, uint8_t *z, ssg_aln_t *out, int *err, unsigned long long *cells, uint8_t *tlds, uint8_t *qlds)
{
	const int rlen = (int)(re - rb);
	int score = 0, n_cigar = 0;
	/* both sequences in LDS when they fit (the usual record): the DP rows and the NM/MD walk read them many times */
	uint8_t *tb = rlen <= SSG_TWIN_LDS ? tlds : tbuf;
	wv_fetch_ref(ix, rb, re, tb);
	for (int i = wv_lane(); i < l_query && i < SSG_ALN_QLDS; i += 64) qlds[i] = query[i];
	ssg_wave_ldssync();
	const uint8_t *qp = l_query <= SSG_ALN_QLDS ? qlds : query;
	const bool rev = rb >= ix.l_pac;
	ssg_seqv_t q = { rev ? qp + l_query - 1 : qp, rev ? -1 : 1 };
	ssg_seqv_t t = { rev ? tb + rlen - 1 : tb, rev ? -1 : 1 };
	if (l_query == rlen && w_ == 0) {
		int sc = 0;
		for (int i = wv_lane(); i < l_query; i += 64) sc += opt.mat[sq_at(t, i) * 5 + sq_at(q, i)];
		score = wv_sum(sc);
		SSG_LANE0(out->cigar[0] = (uint32_t)l_query << 4 | 0);
		n_cigar = 1;
	} else {
		int max_ins = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
		int max_del = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
		int max_gap = max_ins > max_del ? max_ins : max_del;
		max_gap = max_gap > 1 ? max_gap : 1;
		int w = (max_gap + iabs(rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		int min_w = iabs(rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		const long ncol = l_query < 2 * w + 1 ? l_query : 2 * w + 1;
		if (ncol * rlen > SSG_Z_CAP) { *err = 5; return 0; }
		score = wv_global2_any<WIDE>(opt, l_query, q, rlen, t, w, z, cells);
		int nc = 0;
		SSG_LANE0(nc = ssg_global_backtrace(z, l_query, rlen, w, out->cigar, SSG_MAX_CIGAR));
		n_cigar = wv_bcast(nc, 0);
		if (n_cigar > SSG_MAX_CIGAR - 2) { *err = 6; n_cigar = SSG_MAX_CIGAR - 2; }
	}
	/* NM and MD: wave-uniform walk over the CIGAR; match runs are compared 64 bases per step and only the
	 * mismatches (ballot bits) are visited; lane 0 stores the characters */
	int nm = 0, lmd = 0;
	{
		const bool wr = wv_lane() == 0;
		int k, x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, l = 0;
		const char *int2base = rb < ix.l_pac ? "ACGTN" : "TGCAN";
		ssg_wave_memsync();
		for (k = 0; k < n_cigar; ++k) {
			const uint32_t cg = out->cigar[k];
			const int op = cg & 0xf, len = (int)(cg >> 4);
			if (op == 0) {
				for (int i0 = 0; i0 < len; i0 += 64) {
					const int cl = len - i0 < 64 ? len - i0 : 64, i = i0 + wv_lane();
					unsigned long long bal = wv_ballot(i < len && sq_at(q, x + i) != sq_at(t, y + i));
					int prev = 0;
					while (bal) {
						const int bpos = (int)__builtin_ctzll(bal); bal &= bal - 1;
						u += bpos - prev;
						l = ssg_put_int(out->md, l, SSG_MAX_MD - 1, u, wr);
						if (wr && l < SSG_MAX_MD - 1) out->md[l] = int2base[sq_at(t, y + i0 + bpos)];
						++l; ++n_mm; u = 0; prev = bpos + 1;
					}
					u += cl - prev;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < n_cigar - 1) {
					l = ssg_put_int(out->md, l, SSG_MAX_MD - 1, u, wr); if (wr && l < SSG_MAX_MD - 1) out->md[l] = '^'; ++l;
					for (int i = 0; i < len; ++i) { if (wr && l < SSG_MAX_MD - 1) out->md[l] = int2base[sq_at(t, y + i)]; ++l; }
					u = 0; n_gap += len;
				}
				y += len;
			} else if (op == 1) { x += len; n_gap += len; }
		}
		l = ssg_put_int(out->md, l, SSG_MAX_MD - 1, u, wr);
		if (wr) out->md[l < SSG_MAX_MD - 1 ? l : SSG_MAX_MD - 1] = 0;
		nm = n_mm + n_gap; lmd = l;
	}
	if (lmd >= SSG_MAX_MD - 1) *err = 7;
	SSG_LANE0(out->n_cigar = n_cigar; out->NM = nm; out->l_md = lmd);
	return score;
}

/* ---- the banded global alignment with ONE LANE PER REQUEST ----
 * Nine records in ten that need ksw_global2 need it with a narrow band (three or four mismatches: w = 11 or 16; a short indel: w = 4..8).  A wavefront per record
 * runs such a band on a third of its lanes, row after dependent row, then backtracks and walks NM / MD on lane 0: 18 ms of the step for 280 k records.  Here a lane
 * owns a record: the row {H, E} of the band in LDS (a ring of 2w + 2 columns, h:16 | e:16 -- "minus infinity" is -16384, every value of the recurrence is an
 * offset from it or a real score far above it, so all comparisons come out as upstream's), the query as 4-bit codes in LDS, the direction bits four to a cell and
 * eight cells to a word in a per-wave HBM slab ([row][word][lane]: coalesced), then the lane backtracks its own matrix and walks NM / MD from the 2-bit pac.
 * Classes by band width (LDS per wave); a record whose first alignment does not end upstream's loop (score < truesc - a: the band is doubled) is handed to the
 * wave kernel, which starts over. */
#define SSG_R2D_CLASSES 3
#define SSG_R2D_NEG (-16384)
SSG_DEVFN int ssg_r2d_wmax(int cls) { return cls == 0 ? 8 : cls == 1 ? 16 : 32; }
/* the band bwa_gen_cigar2 runs for (w_, l_query, rlen) */
SSG_DEVFN int ssg_gen_cigar_band(const ssg_mem_opt_t &opt, int w_, int l_query, int rlen)
{
	int max_ins = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
	int max_del = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
	int max_gap = max_ins > max_del ? max_ins : max_del;
	max_gap = max_gap > 1 ? max_gap : 1;
	int w = (max_gap + iabs(rlen - l_query) + 1) >> 1;
	w = w < w_ ? w : w_;
	const int min_w = iabs(rlen - l_query) + 3;
	return w > min_w ? w : min_w;
}
SSG_DEVFN int ssg_r2d_class(const ssg_mem_opt_t &opt, int w_, int l_query, int rlen)
{
	if (l_query == rlen && w_ == 0) return -1;   /* (the gap-free branch: never listed) */
	if (l_query < 1 || rlen < 1 || l_query > SSG_ALN_QLDS) return -1;
	const int w = ssg_gen_cigar_band(opt, w_, l_query, rlen);
	return w <= 8 ? 0 : w <= 16 ? 1 : w <= 32 ? 2 : -1;
}

/* grid-strided over list[0 .. *n_list); LDS: (2 wmax + 2 + qwords) x 64 words; zslab: per wave zrows x zw x 64 words */
__global__ void __launch_bounds__(64) ssg_k_reg2aln_dplane(ssg_index_view_t ix, ssg_mem_opt_t opt, const int32_t *list, const unsigned int *n_list, const ssg_alnreq_t *req,
                              const ssg_alnreg_t *regs, const uint8_t *seq, const int64_t *read_off, ssg_aln_t *alns, uint32_t *zslab, int wmax, int qwords, int zw, int zrows,
                              int32_t *err, unsigned long long *cells, int32_t *todo_list, unsigned int *n_todo)
{
#ifdef SSG_EMU
	uint32_t *L = (uint32_t*)emu::dyn_lds;
#else
	extern __shared__ uint32_t ssg_r2d_lds[];
	uint32_t *L = ssg_r2d_lds;
#endif
	const int lane = (int)threadIdx.x;
	const int S = 2 * wmax + 2;
	uint32_t *EH = L + lane, *Q = L + (long)S * 64 + lane;
	uint32_t *Z = zslab + (long)blockIdx.x * zrows * zw * 64 + lane;
	const long n = (long)*n_list;
	const int sa = opt.a, sb = opt.b;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	unsigned long long ncell = 0;
	int myerr = 0;
	for (long t = (long)blockIdx.x * 64 + lane; t < n; t += (long)gridDim.x * 64) {
		const long g = list[t];
		const ssg_alnreq_t rq = req[g];
		const ssg_alnreg_t ar = regs[rq.reg];
		ssg_aln_t *a = alns + g;
		const int l_query = (int)(read_off[rq.read + 1] - read_off[rq.read]);
		const int qb = ar.qb, qe = ar.qe, qlen = qe - qb;
		const int64_t rb = ar.rb, re = ar.re;
		const int tlen = (int)(re - rb);
		int w2;
		{
			const int tmp = ssg_infer_bw(qlen, tlen, ar.truesc, opt.a, opt.o_del, opt.e_del);
			w2 = ssg_infer_bw(qlen, tlen, ar.truesc, opt.a, opt.o_ins, opt.e_ins);
			w2 = w2 > tmp ? w2 : tmp;
			if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
			w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
		}
		const int w = ssg_gen_cigar_band(opt, w2, qlen, tlen);
		const bool rev = rb >= ix.l_pac;
		const uint8_t *query = seq + read_off[rq.read] + qb;
		if (w > wmax || tlen > zrows || (qlen + 7) / 8 > qwords) { todo_list[atomicAdd(n_todo, 1u)] = (int32_t)g; continue; }   /* (not listed for this class by ssg_k_reg2aln_lane) */
		/* query codes in alignment order, eight to a word */
		for (int k = 0; k * 8 < qlen; ++k) {
			uint32_t wd = 0;
			for (int u = 0; u < 8 && k * 8 + u < qlen; ++u) { const int j = k * 8 + u; wd |= (uint32_t)(rev ? query[qlen - 1 - j] : query[j]) << (4 * u); }
			Q[k * 64] = wd;
		}
		/* row -1 */
#define SSG_R2D_PACK(h, e) (((uint32_t)(e) << 16) | ((uint32_t)(h) & 0xffffu))
		EH[0] = SSG_R2D_PACK(0, SSG_R2D_NEG);
		for (int j = 1; j <= qlen && j <= w; ++j) EH[(j % S) * 64] = SSG_R2D_PACK(-(o_ins + e_ins * j), SSG_R2D_NEG);
		if (w + 1 <= qlen) EH[((w + 1) % S) * 64] = SSG_R2D_PACK(SSG_R2D_NEG, SSG_R2D_NEG);   /* (row 0 reads columns 0 .. w only; set for tidiness) */
		const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
		ssg_tgt_t tg;
		ssg_tgt_init(tg, ix, rev ? re - 1 : rb, rev ? -1 : 1);
		int sbeg = 0;   /* ring slot of column beg */
		for (int i = 0; i < tlen; ++i) {
			const int tb = ssg_tgt_next(tg);
			const int beg = i > w ? i - w : 0;
			const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
			if (i > w) { ++sbeg; if (sbeg == S) sbeg = 0; }
			int h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : SSG_R2D_NEG, f = SSG_R2D_NEG;
			int slot = sbeg, zi = 0;
			uint32_t zacc = 0, qw = Q[(beg >> 3) * 64];
			ncell += (unsigned long long)(end - beg);
			for (int j = beg; j < end; ++j) {
				if ((j & 7) == 0) qw = Q[(j >> 3) * 64];
				const uint32_t wd = EH[slot * 64];
				int m = (int)(wd << 16) >> 16, e = (int)wd >> 16, h, tt;
				m += ssg_sc(sa, sb, tb, (int)(qw >> ((j & 7) * 4)) & 15);
				uint32_t d = m >= e ? 0u : 1u;
				h = m >= e ? m : e;
				d = h >= f ? d : 2u;
				h = h >= f ? h : f;
				tt = m - oe_del;
				e -= e_del;
				d |= e > tt ? 4u : 0u;
				e = e > tt ? e : tt;
				EH[slot * 64] = SSG_R2D_PACK(h1, e);
				h1 = h;
				tt = m - oe_ins;
				f -= e_ins;
				d |= f > tt ? 8u : 0u;
				f = f > tt ? f : tt;
				const int c = j - beg;
				zacc |= d << ((c & 7) * 4);
				if ((c & 7) == 7) { Z[((long)i * zw + zi) * 64] = zacc; zacc = 0; ++zi; }
				++slot; if (slot == S) slot = 0;
			}
			if ((end - beg) & 7) Z[((long)i * zw + zi) * 64] = zacc;
			EH[slot * 64] = SSG_R2D_PACK(h1, SSG_R2D_NEG);   /* eh[end] */
		}
		int score;
		{ const uint32_t wd = EH[(qlen % S) * 64]; score = (int)(wd << 16) >> 16; }
		/* upstream's loop ends after this alignment when the score is within one match of the local one, or the band cannot grow */
		if (!(w2 == opt.w << 2 || !(score < ar.truesc - opt.a))) { todo_list[atomicAdd(n_todo, 1u)] = (int32_t)g; continue; }
		/* backtrace (upstream ksw_global2): ops from the end, merged, then reversed */
		int n_cigar = 0;
		{
			int which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
			int cur_op = -1, cur_len = 0, nst = 0;
#define SSG_R2D_PUSH(op, len) do { if ((op) == cur_op) cur_len += (len); else { if (cur_op >= 0) { if (nst < SSG_MAX_CIGAR) a->cigar[nst] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; ++nst; } cur_op = (op); cur_len = (len); } } while (0)
			while (i >= 0 && k >= 0) {
				const int c = k - (i > w ? i - w : 0);
				const uint32_t nib = Z[((long)i * zw + (c >> 3)) * 64] >> ((c & 7) * 4) & 15u;
				which = which == 0 ? (int)(nib & 3u) : which == 1 ? (int)(nib >> 2 & 1u) : (int)(nib >> 3 & 1u) * 2;
				if (which == 0) { SSG_R2D_PUSH(0, 1); --i; --k; }
				else if (which == 1) { SSG_R2D_PUSH(2, 1); --i; }
				else { SSG_R2D_PUSH(1, 1); --k; }
			}
			if (i >= 0) SSG_R2D_PUSH(2, i + 1);
			if (k >= 0) SSG_R2D_PUSH(1, k + 1);
			if (cur_op >= 0) { if (nst < SSG_MAX_CIGAR) a->cigar[nst] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; ++nst; }
#undef SSG_R2D_PUSH
			const int mst = nst < SSG_MAX_CIGAR ? nst : SSG_MAX_CIGAR;
			for (int x = 0; x < mst >> 1; ++x) { const uint32_t t2 = a->cigar[x]; a->cigar[x] = a->cigar[mst - 1 - x]; a->cigar[mst - 1 - x] = t2; }
			n_cigar = nst;
			if (n_cigar > SSG_MAX_CIGAR - 2) { myerr = 6; n_cigar = SSG_MAX_CIGAR - 2; }
			(void)n_col;
		}
		/* NM and MD (upstream bwa_gen_cigar2's walk) */
		int nm, lmd;
		{
			const char *int2base = rev ? "TGCAN" : "ACGTN";
			int x = 0, u = 0, n_mm = 0, n_gap = 0, l = 0;
			ssg_tgt_init(tg, ix, rev ? re - 1 : rb, rev ? -1 : 1);
			for (int k = 0; k < n_cigar; ++k) {
				const uint32_t cg = a->cigar[k];
				const int op = (int)(cg & 0xf), len = (int)(cg >> 4);
				if (op == 0) {
					for (int i = 0; i < len; ++i) {
						const int tb = ssg_tgt_next(tg), qc = (int)(Q[((x + i) >> 3) * 64] >> (((x + i) & 7) * 4)) & 15;
						if (qc != tb) { l = ssg_put_int(a->md, l, SSG_MAX_MD - 1, u); if (l < SSG_MAX_MD - 1) a->md[l] = int2base[tb]; ++l; ++n_mm; u = 0; }
						else ++u;
					}
					x += len;
				} else if (op == 2) {
					const bool mid = k > 0 && k < n_cigar - 1;
					if (mid) { l = ssg_put_int(a->md, l, SSG_MAX_MD - 1, u); if (l < SSG_MAX_MD - 1) a->md[l] = '^'; ++l; }
					for (int i = 0; i < len; ++i) { const int tb = ssg_tgt_next(tg); if (mid) { if (l < SSG_MAX_MD - 1) a->md[l] = int2base[tb]; ++l; } }
					if (mid) { u = 0; n_gap += len; }
				} else if (op == 1) { x += len; n_gap += len; }
			}
			l = ssg_put_int(a->md, l, SSG_MAX_MD - 1, u);
			a->md[l < SSG_MAX_MD - 1 ? l : SSG_MAX_MD - 1] = 0;
			nm = n_mm + n_gap; lmd = l;
			if (lmd >= SSG_MAX_MD - 1) myerr = myerr > 7 ? myerr : 7;
		}
		/* the record (upstream mem_reg2aln after the loop) */
		{
			int is_rev;
			int64_t pos = ssg_depos(ix, rb < ix.l_pac ? rb : re - 1, &is_rev);
			if (n_cigar > 0) { /* squeeze out a leading or trailing deletion */
				if ((a->cigar[0] & 0xf) == 2) { pos += a->cigar[0] >> 4; --n_cigar; for (int k = 0; k < n_cigar; ++k) a->cigar[k] = a->cigar[k+1]; }
				else if ((a->cigar[n_cigar-1] & 0xf) == 2) --n_cigar;
			}
			if (qb != 0 || qe != l_query) {
				const int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
				if (clip5) { for (int k = n_cigar; k > 0; --k) a->cigar[k] = a->cigar[k-1]; a->cigar[0] = (uint32_t)clip5 << 4 | 3; ++n_cigar; }
				if (clip3) a->cigar[n_cigar++] = (uint32_t)clip3 << 4 | 3;
			}
			a->n_cigar = n_cigar; a->NM = nm; a->l_md = lmd;
			a->rid = ssg_pos2rid(ix, pos);
			a->pos = pos - ix.ctg_off[a->rid];
			a->is_rev = is_rev;
			a->flag = rq.flag | (ar.secondary >= 0 ? 0x100 : 0);
			a->mapq = rq.mapq;
			a->score = ar.score; a->sub = ar.sub > ar.csub ? ar.sub : ar.csub;
			a->reg_idx = rq.owner; a->xa_cnt = 0; a->_pad = rq.kind;
		}
	}
#undef SSG_R2D_PACK
	if (cells && ncell) atomicAdd(cells, ncell);
	if (myerr) atomicMax(err, myerr);
}

/* Records whose region aligns without gaps (upstream bwa_gen_cigar2's first branch: equal lengths and a zero band from
 * infer_bw -- the bulk of a batch) need no DP: one LANE per record walks the bases once for NM / MD.  The rest (and any
 * malformed request) goes to the wave-per-record kernel through todo_list. */
__global__ void __launch_bounds__(64) ssg_k_reg2aln_lane(ssg_index_view_t ix, ssg_mem_opt_t opt, long n_req, const ssg_alnreq_t *req, const ssg_alnreg_t *regs,
                              const uint8_t *seq, const int64_t *read_off, ssg_aln_t *alns, int32_t *err, int32_t *todo_list, unsigned int *n_todo,
                              int32_t *dp_list /* [SSG_R2D_CLASSES][n_req] or NULL: requests whose band fits the lane-per-request DP kernel, by class */, unsigned int *n_dp)
{
	const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_req) return;
	const ssg_alnreq_t rq = req[g];
	ssg_aln_t *a = alns + g;
	if (rq.reg < 0) { /* upstream mem_reg2aln(ar == 0) */
		a->pos = -1; a->rid = -1; a->flag = rq.flag | 0x4; a->mapq = 0; a->NM = 0; a->score = 0; a->sub = 0; a->n_cigar = 0;
		a->is_rev = 0; a->l_md = 0; a->reg_idx = -1; a->xa_cnt = 0; a->md[0] = 0;
		return;
	}
	const ssg_alnreg_t ar = regs[rq.reg];
	const int l_query = (int)(read_off[rq.read + 1] - read_off[rq.read]);
	const int qb = ar.qb, qe = ar.qe, lq = qe - qb;
	const int64_t rb = ar.rb, re = ar.re;
	int w2 = -1;
	if (!(re - rb > SSG_TWIN_GLB || rb < 0 || re > ix.l_pac << 1 || rb >= re || (rb < ix.l_pac && re > ix.l_pac))) {
		const int tmp = ssg_infer_bw(lq, (int)(re - rb), ar.truesc, opt.a, opt.o_del, opt.e_del);
		w2 = ssg_infer_bw(lq, (int)(re - rb), ar.truesc, opt.a, opt.o_ins, opt.e_ins);
		w2 = w2 > tmp ? w2 : tmp;
		if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
	}
	if (w2 != 0 || (int64_t)lq != re - rb) {
		if (SSG_TUNING) { const int wb = w2 < 0 ? 15 : w2 == 0 ? 0 : w2 <= 4 ? 1 : w2 <= 8 ? 2 : w2 <= 12 ? 3 : w2 <= 16 ? 4 : w2 <= 24 ? 5 : w2 <= 32 ? 6 : w2 <= 48 ? 7 : w2 <= 64 ? 8 : w2 < 100 ? 9 : 10; atomicAdd(&ssg_dbg_cyc[48 + wb], 1ull); }
		if (dp_list && w2 >= 0) {
			const int cls = ssg_r2d_class(opt, w2 < opt.w << 2 ? w2 : opt.w << 2, lq, (int)(re - rb));
			if (cls >= 0) { dp_list[(long)cls * n_req + atomicAdd(&n_dp[cls], 1u)] = (int32_t)g; return; }
		}
		todo_list[atomicAdd(n_todo, 1u)] = (int32_t)g; return;
	}
	const uint8_t *query = seq + read_off[rq.read] + qb;
	const bool rev = rb >= ix.l_pac;
	const char *int2base = rev ? "TGCAN" : "ACGTN";
	int u = 0, n_mm = 0, l = 0;
	ssg_tgt_t tg;   /* reference bases from the 2-bit pac, 16 per aligned word (k_extlane.h) */
	ssg_tgt_init(tg, ix, rev ? re - 1 : rb, rev ? -1 : 1);
	uint64_t q8 = 0;   /* eight bases of the read at a time: a byte per trip of this loop fetched the read's line again and again (the wave's other lanes' lines push it out of the cache in between: 6 KB fetched per record, PMC round 5) */
	for (int i = 0; i < lq; ++i) {
		if ((i & 7) == 0) {
			q8 = 0;
			SSG_UNROLL for (int b = 0; b < 8; ++b) if (i + b < lq) q8 |= (uint64_t)(rev ? query[lq - 1 - i - b] : query[i + b]) << (8 * b);
		}
		const int tb = ssg_tgt_next(tg);
		const int qc = (int)(q8 >> ((i & 7) << 3)) & 0xff;
		if (qc != tb) { l = ssg_put_int(a->md, l, SSG_MAX_MD - 1, u); if (l < SSG_MAX_MD - 1) a->md[l] = int2base[tb]; ++l; ++n_mm; u = 0; }
		else ++u;
	}
	l = ssg_put_int(a->md, l, SSG_MAX_MD - 1, u);
	a->md[l < SSG_MAX_MD - 1 ? l : SSG_MAX_MD - 1] = 0;
	if (l >= SSG_MAX_MD - 1) atomicMax(err, 7);
	int n_cigar = 1, is_rev;
	a->cigar[0] = (uint32_t)lq << 4 | 0;
	const int64_t pos = ssg_depos(ix, rb < ix.l_pac ? rb : re - 1, &is_rev);
	if (qb != 0 || qe != l_query) {
		const int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
		if (clip5) { a->cigar[1] = a->cigar[0]; a->cigar[0] = (uint32_t)clip5 << 4 | 3; ++n_cigar; }
		if (clip3) a->cigar[n_cigar++] = (uint32_t)clip3 << 4 | 3;
	}
	a->n_cigar = n_cigar; a->NM = n_mm; a->l_md = l;
	a->rid = ssg_pos2rid(ix, pos);
	a->pos = pos - ix.ctg_off[a->rid];
	a->is_rev = is_rev;
	a->flag = rq.flag | (ar.secondary >= 0 ? 0x100 : 0);
	a->mapq = rq.mapq;
	a->score = ar.score; a->sub = ar.sub > ar.csub ? ar.sub : ar.csub;
	a->reg_idx = rq.owner; a->xa_cnt = 0; a->_pad = rq.kind;
}

#ifndef SSG_R2A_WAVES
#define SSG_R2A_WAVES 4   /* 128 VGPRs: 28 ms against 40 at 2 waves/SIMD (211 VGPRs) */
#endif
template <bool WIDE>
__global__ void __launch_bounds__(256, SSG_R2A_WAVES) ssg_k_reg2aln(ssg_index_view_t ix, ssg_mem_opt_t opt, long n_req, const ssg_alnreq_t *req, const ssg_alnreg_t *regs,
                              const uint8_t *seq, const int64_t *read_off, ssg_aln_t *alns, uint8_t *tglb, uint8_t *zglb, int32_t *err, unsigned long long *cells,
                              const int32_t *todo_list, const unsigned int *n_todo /* the records ssg_k_reg2aln_lane left (NULL: all n_req) */)
{
	__shared__ uint8_t tlds_[SSG_WAVES_PER_WG][SSG_TWIN_LDS], qlds_[SSG_WAVES_PER_WG][SSG_ALN_QLDS];
	const int wslot = (int)(threadIdx.x >> 6);
	const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + wslot, nwaves = (long)gridDim.x * (blockDim.x >> 6);
	uint8_t *tg = tglb + wave0 * (long)SSG_TWIN_GLB, *z = zglb + wave0 * (long)SSG_Z_CAP;
	unsigned long long nc = 0;
	int myerr = 0;
	const long n_work = todo_list ? (long)*n_todo : n_req;
	for (long gw = wave0; gw < n_work; gw += nwaves) {
		const long g = todo_list ? todo_list[gw] : gw;
		const ssg_alnreq_t rq = req[g];
		ssg_aln_t *a = alns + g;
		if (rq.reg < 0) { /* upstream mem_reg2aln(ar == 0) */
			SSG_LANE0(a->pos = -1; a->rid = -1; a->flag = rq.flag | 0x4; a->mapq = 0; a->NM = 0; a->score = 0; a->sub = 0; a->n_cigar = 0;
			          a->is_rev = 0; a->l_md = 0; a->reg_idx = -1; a->xa_cnt = 0; a->md[0] = 0);
			continue;
		}
		const ssg_alnreg_t ar = regs[rq.reg];
		const uint8_t *query = seq + read_off[rq.read];
		const int l_query = (int)(read_off[rq.read + 1] - read_off[rq.read]);
		int i, w2, tmp, qb = ar.qb, qe = ar.qe, score = 0, last_sc = -(1 << 30), is_rev;
		int64_t rb = ar.rb, re = ar.re, pos;
		if (re - rb > SSG_TWIN_GLB || rb < 0 || re > ix.l_pac << 1 || rb >= re || (rb < ix.l_pac && re > ix.l_pac)) { myerr = 8; continue; }
		tmp = ssg_infer_bw(qe - qb, (int)(re - rb), ar.truesc, opt.a, opt.o_del, opt.e_del);
		w2  = ssg_infer_bw(qe - qb, (int)(re - rb), ar.truesc, opt.a, opt.o_ins, opt.e_ins);
		w2 = w2 > tmp ? w2 : tmp;
		if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
		i = 0;
		do {
			w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
			score = wv_gen_cigar<WIDE>(ix, opt, w2, qe - qb, query + qb, rb, re, tg, z, a, &myerr, &nc, tlds_[wslot], qlds_[wslot]);
			if (score == last_sc || w2 == opt.w << 2) break;
			last_sc = score;
			w2 <<= 1;
		} while (++i < 3 && score < ar.truesc - opt.a);
		pos = ssg_depos(ix, rb < ix.l_pac ? rb : re - 1, &is_rev);
		SSG_LANE0(
			int n_cigar = a->n_cigar;
			if (n_cigar > 0) { /* squeeze out a leading or trailing deletion */
				if ((a->cigar[0] & 0xf) == 2) { pos += a->cigar[0] >> 4; --n_cigar; for (int k = 0; k < n_cigar; ++k) a->cigar[k] = a->cigar[k+1]; }
				else if ((a->cigar[n_cigar-1] & 0xf) == 2) --n_cigar;
			}
			if (qb != 0 || qe != l_query) {
				int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
				if (clip5) { for (int k = n_cigar; k > 0; --k) a->cigar[k] = a->cigar[k-1]; a->cigar[0] = (uint32_t)clip5 << 4 | 3; ++n_cigar; }
				if (clip3) a->cigar[n_cigar++] = (uint32_t)clip3 << 4 | 3;
			}
			a->n_cigar = n_cigar;
			a->rid = ssg_pos2rid(ix, pos);
			a->pos = pos - ix.ctg_off[a->rid];
			a->is_rev = is_rev;
			a->flag = rq.flag | (ar.secondary >= 0 ? 0x100 : 0);
			a->mapq = rq.mapq;
			a->score = ar.score; a->sub = ar.sub > ar.csub ? ar.sub : ar.csub;
			a->reg_idx = rq.owner; a->xa_cnt = 0; a->_pad = rq.kind);
	}
	if (wv_lane() == 0) { if (cells) atomicAdd(cells, nc); if (myerr) atomicMax(err, myerr); }
}
#endif
