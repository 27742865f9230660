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

/* Records whose region aligns without gaps (upstream bwa_gen_cigar2's first branch: equal lengths and a zero band from
 * infer_bw -- the bulk of a batch) need no DP: one LANE per record walks the bases once for NM / MD.  The rest (and any
 * malformed request) goes to the wave-per-record kernel through todo_list. */
__global__ void __launch_bounds__(64) ssg_k_reg2aln_lane(ssg_index_view_t ix, ssg_mem_opt_t opt, long n_req, const ssg_alnreq_t *req, const ssg_alnreg_t *regs,
                              const uint8_t *seq, const int64_t *read_off, ssg_aln_t *alns, int32_t *err, int32_t *todo_list, unsigned int *n_todo)
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
	if (w2 != 0 || (int64_t)lq != re - rb) { todo_list[atomicAdd(n_todo, 1u)] = (int32_t)g; return; }
	const uint8_t *query = seq + read_off[rq.read] + qb;
	const bool rev = rb >= ix.l_pac;
	const char *int2base = rev ? "TGCAN" : "ACGTN";
	int u = 0, n_mm = 0, l = 0;
	ssg_tgt_t tg;   /* reference bases from the 2-bit pac, 16 per aligned word (k_extlane.h) */
	ssg_tgt_init(tg, ix, rev ? re - 1 : rb, rev ? -1 : 1);
	for (int i = 0; i < lq; ++i) {
		const int tb = ssg_tgt_next(tg);
		const int qc = rev ? query[lq - 1 - i] : query[i];
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
