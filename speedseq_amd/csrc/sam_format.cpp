/*
 * sam_format.cpp -- host side of the hot path's output boundary: SAM text from the device-produced
 * alignment records (upstream mem_aln2sam / mem_reg2sam's printing half and mem_gen_alt's XA
 * strings, bwamem.c / bwamem_extra.c).  Pure text assembly: every number it
 * prints was computed on the GPU.
 */
#include <string>
#include <vector>
#include <thread>
#include <string.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"
SSG_ABI_FP_DEFINE(sam_format)

namespace {
struct alignas(128) sbuf {   /* one per formatting thread, side by side in a vector: its length field is written on every append, so it gets cache lines of its own */
	std::string s;
	void putl(long long v)
	{	/* decimal, by hand: a SAM line prints a dozen integers and snprintf was a third of the formatter's time */
		char b[24]; int n = 24; unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
		do { b[--n] = (char)('0' + u % 10); u /= 10; } while (u);
		if (v < 0) b[--n] = '-';
		s.append(b + n, (size_t)(24 - n));
	}
	void putc(char c) { s.push_back(c); }
	void puts(const char *p) { s.append(p); }
	char *grow(size_t n) { const size_t o = s.size(); s.resize(o + n); return &s[o]; }   /* n bytes to be written by the caller */
};

inline int get_rlen(int n_cigar, const uint32_t *cigar)
{
	int l = 0;
	for (int k = 0; k < n_cigar; ++k) { int op = cigar[k] & 0xf; if (op == 0 || op == 2) l += cigar[k] >> 4; }
	return l;
}

/* the mate as mem_aln2sam sees it (a private copy it may rewrite) */
struct mate_t { int rid; long long pos; int is_rev, n_cigar; const uint32_t *cigar; };

void aln2sam(const ssg_index_t *idx, sbuf &str, const char *name, int l_seq, const uint8_t *seq, const char *qual, const char *comment,
             int n, const ssg_aln_t *const *list, int which, const mate_t *m_, const std::string *XA, const char *rg_id, bool softclip)
{	/* upstream mem_aln2sam */
	const ssg_aln_t &a = *list[which];
	int flag = a.flag, rid = a.rid, is_rev = a.is_rev, n_cigar = a.n_cigar; long long pos = a.pos;
	mate_t mt, *m = 0;
	if (m_) { mt = *m_; m = &mt; }
	flag |= m ? 0x1 : 0;
	flag |= rid < 0 ? 0x4 : 0;
	flag |= m && m->rid < 0 ? 0x8 : 0;
	if (rid < 0 && m && m->rid >= 0) { rid = m->rid; pos = m->pos; is_rev = m->is_rev; n_cigar = 0; }
	if (m && m->rid < 0 && rid >= 0) { m->rid = rid; m->pos = pos; m->is_rev = is_rev; m->n_cigar = 0; }
	flag |= is_rev ? 0x10 : 0;
	flag |= m && m->is_rev ? 0x20 : 0;
	str.puts(name); str.putc('\t');
	str.putl((flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0)); str.putc('\t');
	if (rid >= 0) {
		str.puts(ssg_index_name(idx, rid)); str.putc('\t');
		str.putl(pos + 1); str.putc('\t');
		str.putl(a.mapq); str.putc('\t');
		if (n_cigar) {
			for (int i = 0; i < n_cigar; ++i) {
				int c = a.cigar[i] & 0xf;
				if (!softclip && (c == 3 || c == 4)) c = which ? 4 : 3;
				str.putl(a.cigar[i] >> 4); str.putc("MIDSH"[c]);
			}
		} else str.putc('*');
	} else str.puts("*\t0\t0\t*");
	str.putc('\t');
	if (m && m->rid >= 0) {
		if (rid == m->rid) str.putc('='); else str.puts(ssg_index_name(idx, m->rid));
		str.putc('\t');
		str.putl(m->pos + 1); str.putc('\t');
		if (rid == m->rid) {
			long long p0 = pos + (is_rev ? get_rlen(n_cigar, a.cigar) - 1 : 0);
			long long p1 = m->pos + (m->is_rev ? get_rlen(m->n_cigar, m->cigar) - 1 : 0);
			if (m->n_cigar == 0 || n_cigar == 0) str.putc('0');
			else str.putl(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else str.putc('0');
	} else str.puts("*\t0\t0");
	str.putc('\t');
	if (flag & 0x100) str.puts("*\t*");
	else if (!is_rev) {
		int qb = 0, qe = l_seq;
		if (n_cigar && which && !softclip) {
			if ((a.cigar[0] & 0xf) == 4 || (a.cigar[0] & 0xf) == 3) qb += a.cigar[0] >> 4;
			if ((a.cigar[n_cigar-1] & 0xf) == 4 || (a.cigar[n_cigar-1] & 0xf) == 3) qe -= a.cigar[n_cigar-1] >> 4;
		}
		{ char *d = str.grow((size_t)(qe > qb ? qe - qb : 0)); for (int i = qb; i < qe; ++i) *d++ = "ACGTN"[seq[i]]; }
		str.putc('\t');
		if (qual) str.s.append(qual + qb, qe - qb); else str.putc('*');
	} else {
		int qb = 0, qe = l_seq;
		if (n_cigar && which && !softclip) {
			if ((a.cigar[0] & 0xf) == 4 || (a.cigar[0] & 0xf) == 3) qe -= a.cigar[0] >> 4;
			if ((a.cigar[n_cigar-1] & 0xf) == 4 || (a.cigar[n_cigar-1] & 0xf) == 3) qb += a.cigar[n_cigar-1] >> 4;
		}
		{ char *d = str.grow((size_t)(qe > qb ? qe - qb : 0)); for (int i = qe - 1; i >= qb; --i) *d++ = "TGCAN"[seq[i]]; }
		str.putc('\t');
		if (qual) { char *d = str.grow((size_t)(qe > qb ? qe - qb : 0)); for (int i = qe - 1; i >= qb; --i) *d++ = qual[i]; } else str.putc('*');
	}
	if (n_cigar) {
		str.puts("\tNM:i:"); str.putl(a.NM);
		str.puts("\tMD:Z:"); str.s.append(a.md, a.l_md);
	}
	if (a.score >= 0) { str.puts("\tAS:i:"); str.putl(a.score); }
	if (a.sub >= 0) { str.puts("\tXS:i:"); str.putl(a.sub); }
	if (rg_id && rg_id[0]) { str.puts("\tRG:Z:"); str.puts(rg_id); }
	if (!(flag & 0x100)) {
		int i;
		for (i = 0; i < n; ++i) if (i != which && !(list[i]->flag & 0x100)) break;
		if (i < n) {
			str.puts("\tSA:Z:");
			for (i = 0; i < n; ++i) {
				const ssg_aln_t &r = *list[i];
				if (i == which || (r.flag & 0x100)) continue;
				str.puts(ssg_index_name(idx, r.rid)); str.putc(',');
				str.putl(r.pos + 1); str.putc(',');
				str.putc("+-"[r.is_rev]); str.putc(',');
				for (int k = 0; k < r.n_cigar; ++k) { str.putl(r.cigar[k] >> 4); str.putc("MIDSH"[r.cigar[k] & 0xf]); }
				str.putc(','); str.putl(r.mapq);
				str.putc(','); str.putl(r.NM);
				str.putc(';');
			}
		}
	}
	if (XA && !XA->empty()) { str.puts("\tXA:Z:"); str.s.append(*XA); }
	if (comment) { str.putc('\t'); str.puts(comment); }
	str.putc('\n');
}

/* ---- the same record as the bytes `sambamba view -S -f bam` makes of aln2sam's line (htslib sam.c:835-1028 sam_parse1 rules and
 * sam.c:443-473 bam_write1 layout): block_size, refID, pos, bin<<16|mapq<<8|l_qname, flag<<16|n_cigar, l_seq, mate refID / pos, tlen,
 * qname, cigar, 4-bit seq, qual, aux with the smallest integer types.  Field values follow aln2sam line for line. ---- */
inline int reg2bin(int64_t beg, int64_t end)
{	/* hts_reg2bin(beg, end, 14, 5), hts.h:580-586 */
	int l, s = 14, t = ((1 << 15) - 1) / 7;
	for (--end, l = 5; l > 0; --l, s += 3, t -= 1 << ((l << 1) + l)) if (beg >> s == end >> s) return t + (int)(beg >> s);
	return 0;
}
struct alignas(128) bbuf {
	std::vector<uint8_t> b;
	void put(const void *p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
	void tag_int(const char *t, long long v)
	{	/* sam_parse1's integer typing (sam.c:964-988) */
		b.push_back((uint8_t)t[0]); b.push_back((uint8_t)t[1]);
		if (v < 0) {
			if (v >= INT8_MIN) { b.push_back('c'); b.push_back((uint8_t)(int8_t)v); }
			else if (v >= INT16_MIN) { int16_t y = (int16_t)v; b.push_back('s'); put(&y, 2); }
			else { int32_t y = (int32_t)v; b.push_back('i'); put(&y, 4); }
		} else {
			if (v <= UINT8_MAX) { b.push_back('C'); b.push_back((uint8_t)v); }
			else if (v <= UINT16_MAX) { uint16_t y = (uint16_t)v; b.push_back('S'); put(&y, 2); }
			else { uint32_t y = (uint32_t)v; b.push_back('I'); put(&y, 4); }
		}
	}
	void tag_str(const char *t, const char *v, size_t n) { b.push_back((uint8_t)t[0]); b.push_back((uint8_t)t[1]); b.push_back('Z'); put(v, n); b.push_back(0); }
};

void aln2bam(const ssg_index_t *idx, bbuf &out, const char *name, int l_seq, const uint8_t *seq, const char *qual,
             int n, const ssg_aln_t *const *list, int which, const mate_t *m_, const std::string *XA, const char *rg_id, bool softclip)
{
	const ssg_aln_t &a = *list[which];
	int flag = a.flag, rid = a.rid, is_rev = a.is_rev, n_cigar = a.n_cigar; long long pos = a.pos;
	mate_t mt, *m = 0;
	if (m_) { mt = *m_; m = &mt; }
	flag |= m ? 0x1 : 0;
	flag |= rid < 0 ? 0x4 : 0;
	flag |= m && m->rid < 0 ? 0x8 : 0;
	if (rid < 0 && m && m->rid >= 0) { rid = m->rid; pos = m->pos; is_rev = m->is_rev; n_cigar = 0; }
	if (m && m->rid < 0 && rid >= 0) { m->rid = rid; m->pos = pos; m->is_rev = is_rev; m->n_cigar = 0; }
	flag |= is_rev ? 0x10 : 0;
	flag |= m && m->is_rev ? 0x20 : 0;
	/* flag 0x10000 (upstream -M: a supplementary line shown as secondary) becomes 0x100 only in the stored FLAG; the SEQ / QUAL / SA decisions below
	 * test the raw flag, as mem_aln2sam and the text path (aln2sam above) do */
	const size_t base = out.b.size();
	out.b.resize(base + 36);
	const size_t l_qname = strlen(name) + 1;
	out.put(name, l_qname);
	int32_t tid = -1, bpos = -1, mapq = 0; int64_t rl = 0;
	if (rid >= 0) {
		tid = rid; bpos = (int32_t)pos; mapq = a.mapq;
		for (int i = 0; i < n_cigar; ++i) {
			int c = a.cigar[i] & 0xf;
			if (!softclip && (c == 3 || c == 4)) c = which ? 4 : 3;
			const uint32_t len = a.cigar[i] >> 4, v = len << 4 | (uint32_t)(c <= 2 ? c : c + 1);   /* "MIDSH" -> BAM codes M0 I1 D2 S4 H5 */
			out.put(&v, 4);
			if (c == 0 || c == 2) rl += len;
		}
		if (!n_cigar) flag |= 4;            /* sam_parse1: a record without CIGAR is treated as unmapped */
	} else { n_cigar = 0; flag |= 4; }
	const int64_t rlen = (!(flag & 4) && n_cigar) ? rl : 1;
	const int bin = reg2bin(bpos, bpos + rlen);
	int32_t mtid = -1, mpos = -1, isize = 0;
	if (m && m->rid >= 0) {
		mtid = m->rid; mpos = (int32_t)m->pos;
		if (rid == m->rid) {
			long long p0 = pos + (is_rev ? get_rlen(n_cigar, a.cigar) - 1 : 0);
			long long p1 = m->pos + (m->is_rev ? get_rlen(m->n_cigar, m->cigar) - 1 : 0);
			if (!(m->n_cigar == 0 || n_cigar == 0)) isize = (int32_t)(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		}
	}
	int32_t l_qseq = 0;
	if (!(flag & 0x100)) {
		int qb = 0, qe = l_seq;
		const bool cl0 = n_cigar && which && !softclip && ((a.cigar[0] & 0xf) == 4 || (a.cigar[0] & 0xf) == 3), cl1 = n_cigar && which && !softclip && ((a.cigar[n_cigar-1] & 0xf) == 4 || (a.cigar[n_cigar-1] & 0xf) == 3);
		if (!is_rev) { if (cl0) qb += a.cigar[0] >> 4; if (cl1) qe -= a.cigar[n_cigar-1] >> 4; }
		else { if (cl0) qe -= a.cigar[0] >> 4; if (cl1) qb += a.cigar[n_cigar-1] >> 4; }
		l_qseq = qe > qb ? qe - qb : 0;
		static const uint8_t fw[5] = { 1, 2, 4, 8, 15 }, rv[5] = { 8, 4, 2, 1, 15 };
		const size_t o = out.b.size();
		out.b.resize(o + (size_t)((l_qseq + 1) >> 1) + (size_t)l_qseq, 0);
		uint8_t *ps = out.b.data() + o, *pq = ps + ((l_qseq + 1) >> 1);
		if (!is_rev) for (int i = 0; i < l_qseq; ++i) ps[i >> 1] |= (uint8_t)(fw[seq[qb + i]] << ((~i & 1) << 2));
		else for (int i = 0; i < l_qseq; ++i) ps[i >> 1] |= (uint8_t)(rv[seq[qe - 1 - i]] << ((~i & 1) << 2));
		if (!qual) memset(pq, 0xff, (size_t)l_qseq);
		else if (!is_rev) for (int i = 0; i < l_qseq; ++i) pq[i] = (uint8_t)(qual[qb + i] - 33);
		else for (int i = 0; i < l_qseq; ++i) pq[i] = (uint8_t)(qual[qe - 1 - i] - 33);
	}
	if (n_cigar) { out.tag_int("NM", a.NM); out.tag_str("MD", a.md, (size_t)a.l_md); }
	if (a.score >= 0) out.tag_int("AS", a.score);
	if (a.sub >= 0) out.tag_int("XS", a.sub);
	if (rg_id && rg_id[0]) out.tag_str("RG", rg_id, strlen(rg_id));
	if (!(flag & 0x100)) {
		int i;
		for (i = 0; i < n; ++i) if (i != which && !(list[i]->flag & 0x100)) break;
		if (i < n) {
			sbuf sa;
			for (i = 0; i < n; ++i) {
				const ssg_aln_t &r = *list[i];
				if (i == which || (r.flag & 0x100)) continue;
				sa.puts(ssg_index_name(idx, r.rid)); sa.putc(',');
				sa.putl(r.pos + 1); sa.putc(',');
				sa.putc("+-"[r.is_rev]); sa.putc(',');
				for (int k = 0; k < r.n_cigar; ++k) { sa.putl(r.cigar[k] >> 4); sa.putc("MIDSH"[r.cigar[k] & 0xf]); }
				sa.putc(','); sa.putl(r.mapq);
				sa.putc(','); sa.putl(r.NM);
				sa.putc(';');
			}
			out.tag_str("SA", sa.s.data(), sa.s.size());
		}
	}
	if (XA && !XA->empty()) out.tag_str("XA", XA->data(), XA->size());
	uint32_t x[9];
	x[0] = (uint32_t)(out.b.size() - base - 4);
	x[1] = (uint32_t)tid; x[2] = (uint32_t)bpos; x[3] = (uint32_t)bin << 16 | (uint32_t)(mapq & 0xff) << 8 | (uint32_t)l_qname;
	x[4] = (uint32_t)((flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0)) << 16 | (uint32_t)n_cigar; x[5] = (uint32_t)l_qseq; x[6] = (uint32_t)mtid; x[7] = (uint32_t)mpos; x[8] = (uint32_t)isize;
	memcpy(out.b.data() + base, x, 36);
}
} // namespace

/* pairs sel[p0..p1) (or p0..p1 themselves when sel == NULL) into `out` (text) or `bout` (BAM records); offs entries relative to the buffer's start */
static int format_range(const ssg_index_t *idx, const ssg_pe_result_t *res, const int32_t *sel, int p0, int p1, const char *const *names, const uint8_t *seq, const int64_t *off,
                        const char *const *quals, const char *const *comments, const char *rg_id, bool softclip, bool se, sbuf *out, bbuf *bout, int64_t *offs)
{
	const int64_t *req_off = ssg_pe_req_off(res);
	const ssg_alnreq_t *req = ssg_pe_req(res);
	const ssg_aln_t *alns = ssg_pe_alns(res);
	std::vector<const ssg_aln_t*> mains[2];
	std::vector<std::string> xa[2];
	std::vector<int> owner;
	if (se) {   /* single-end reads (ssg_mem_process_reads): the units are reads; upstream mem_reg2sam with no mate */
		for (int q = p0; q < p1; ++q) {
			const int r = sel ? sel[q] : q;
			mains[0].clear(); xa[0].clear(); owner.clear();
			for (int64_t g = req_off[r]; g < req_off[r+1]; ++g)
				if (req[g].kind == SSG_REQ_MAIN) { mains[0].push_back(&alns[g]); owner.push_back(req[g].owner); xa[0].emplace_back(); }
			for (int64_t g = req_off[r]; g < req_off[r+1]; ++g) {
				if (req[g].kind != SSG_REQ_XA) continue;
				const ssg_aln_t &t = alns[g];
				for (size_t k = 0; k < owner.size(); ++k) {
					if (owner[k] != req[g].owner || mains[0][k]->rid < 0) continue;
					sbuf x;
					x.puts(ssg_index_name(idx, t.rid)); x.putc(','); x.putc("+-"[t.is_rev]); x.putl(t.pos + 1); x.putc(',');
					for (int c = 0; c < t.n_cigar; ++c) { x.putl(t.cigar[c] >> 4); x.putc("MIDSHN"[t.cigar[c] & 0xf]); }
					x.putc(','); x.putl(t.NM); x.putc(';');
					xa[0][k] += x.s;
				}
			}
			if (mains[0].empty()) return SSG_EINVAL;
			offs[q] = (int64_t)(out ? out->s.size() : bout->b.size());
			const int l_seq = (int)(off[r+1] - off[r]);
			for (size_t k = 0; k < mains[0].size(); ++k) {
				if (out) aln2sam(idx, *out, names[r], l_seq, seq + off[r], quals ? quals[r] : 0, comments ? comments[r] : 0, (int)mains[0].size(), mains[0].data(), (int)k, 0, &xa[0][k], rg_id, softclip);
				else aln2bam(idx, *bout, names[r], l_seq, seq + off[r], quals ? quals[r] : 0, (int)mains[0].size(), mains[0].data(), (int)k, 0, &xa[0][k], rg_id, softclip);
			}
		}
		return 0;
	}
	for (int q = p0; q < p1; ++q) {
		const int p = sel ? sel[q] : q;
		mate_t mate[2];
		for (int i = 0; i < 2; ++i) {
			const int r = 2 * p + i;
			mains[i].clear(); xa[i].clear(); owner.clear();
			for (int64_t g = req_off[r]; g < req_off[r+1]; ++g) {
				if (req[g].kind == SSG_REQ_MAIN) { mains[i].push_back(&alns[g]); owner.push_back(req[g].owner); xa[i].emplace_back(); }
			}
			for (int64_t g = req_off[r]; g < req_off[r+1]; ++g) {
				if (req[g].kind != SSG_REQ_XA) continue;
				const ssg_aln_t &t = alns[g];
				for (size_t k = 0; k < owner.size(); ++k) {
					if (owner[k] != req[g].owner || mains[i][k]->rid < 0) continue;
					sbuf x;
					x.puts(ssg_index_name(idx, t.rid)); x.putc(','); x.putc("+-"[t.is_rev]); x.putl(t.pos + 1); x.putc(',');
					for (int c = 0; c < t.n_cigar; ++c) { x.putl(t.cigar[c] >> 4); x.putc("MIDSHN"[t.cigar[c] & 0xf]); }
					x.putc(','); x.putl(t.NM); x.putc(';');
					xa[i][k] += x.s;
				}
			}
			if (mains[i].empty()) return SSG_EINVAL;
			const ssg_aln_t &h = *mains[i][0];   /* the mate record the other end prints against */
			mate[i].rid = h.rid; mate[i].pos = h.pos; mate[i].is_rev = h.is_rev; mate[i].n_cigar = h.n_cigar; mate[i].cigar = h.cigar;
		}
		for (int i = 0; i < 2; ++i) {
			const int r = 2 * p + i;
			offs[2 * q + i] = (int64_t)(out ? out->s.size() : bout->b.size());
			const int l_seq = (int)(off[r+1] - off[r]);
			for (size_t k = 0; k < mains[i].size(); ++k) {
				if (out) aln2sam(idx, *out, names[r], l_seq, seq + off[r], quals ? quals[r] : 0, comments ? comments[r] : 0,
				                 (int)mains[i].size(), mains[i].data(), (int)k, &mate[!i], &xa[i][k], rg_id, softclip);
				else aln2bam(idx, *bout, names[r], l_seq, seq + off[r], quals ? quals[r] : 0,
				             (int)mains[i].size(), mains[i].data(), (int)k, &mate[!i], &xa[i][k], rg_id, softclip);
			}
		}
	}
	return 0;
}

/* Assembly is split over host threads by ranges of pairs (opt->n_threads, as upstream's worker2 threads print), then joined in input order. */
template <class BUF, class GET>
static int format_all(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, const int32_t *sel, int n_pairs,
                      const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals, const char *const *comments,
                      const char *rg_id, std::vector<BUF> &outs, GET bytes_of, bool text, char **outp, int64_t *offs, bool se = false)
{
	const int per = se ? 1 : 2;   /* reads per unit: n_pairs counts reads when the result is single-end */
	if ((ssg_pe_is_se(res) != 0) != se) return SSG_EINVAL;   /* a single-end result is printed by ssg_sam_format_se, a paired one by the others */
	int T = opt && opt->n_threads > 1 ? opt->n_threads : 1;
	{ const int spare = (int)std::min(48u, std::thread::hardware_concurrency() / 4); if (T > 1 && spare > T) T = spare; }   /* -t sizes upstream's batches; printing has to keep up with an MI355X, not with -t CPU aligners */
	{ const char *e = getenv("SSG_FMT_THREADS"); if (e && atoi(e) > 0) T = atoi(e); }
	if (T > 64) T = 64;
	if (T > (n_pairs + 1023) / 1024) T = (n_pairs + 1023) / 1024;
	if (T < 1) T = 1;
	if ((int)outs.size() < T) outs.resize(T);
	std::vector<int> rcs(T, 0); std::vector<std::thread> th;
	auto lo = [&](int t) { return (int)((int64_t)n_pairs * t / T); };
	for (int t = 0; t < T; ++t) {
		auto work = [&, t]() {
			auto &B = bytes_of(outs[t]); B.clear(); B.reserve((size_t)(lo(t + 1) - lo(t)) * (text ? 1000 : 800));
			sbuf *so = 0; bbuf *bo = 0;
			if constexpr (std::is_same<BUF, sbuf>::value) so = &outs[t]; else bo = &outs[t];
			rcs[t] = format_range(idx, res, sel, lo(t), lo(t + 1), names, seq, off, quals, comments, rg_id, opt && (opt->flag & SSG_F_SOFTCLIP), se, so, bo, offs); };   /* -Y: upstream MEM_F_SOFTCLIP */
		if (T == 1) work(); else th.emplace_back(work);
	}
	for (auto &x : th) x.join();
	th.clear();
	size_t tot = 0; std::vector<size_t> base(T + 1, 0);
	for (int t = 0; t < T; ++t) { if (rcs[t]) return rcs[t]; base[t] = tot; tot += bytes_of(outs[t]).size(); }
	char *buf = (char*)malloc(tot + 1);
	if (!buf) return SSG_ENOMEM;
	for (int t = 0; t < T; ++t) {   /* joined in input order, each part copied by its own thread */
		auto work = [&, t]() { memcpy(buf + base[t], bytes_of(outs[t]).data(), bytes_of(outs[t]).size()); for (int r = per * lo(t); r < per * lo(t + 1); ++r) offs[r] += (int64_t)base[t]; };
		if (T == 1) work(); else th.emplace_back(work);
	}
	for (auto &x : th) x.join();
	buf[tot] = 0; offs[per * n_pairs] = (int64_t)tot;
	*outp = buf;
	return 0;
}

/* the per-thread buffers persist across calls of the calling thread (bwa's formatter thread): after the first batch they are warm
 * memory instead of hundreds of MB of fresh pages per call */
extern "C" int ssg_sam_format_sel(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, const int32_t *sel, int n_sel,
                                  const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals, const char *const *comments,
                                  const char *rg_id, char **sam, int64_t *sam_off)
{
	static thread_local std::vector<sbuf> keep;
	return format_all(idx, opt, res, sel, n_sel, names, seq, off, quals, comments, rg_id, keep, [](sbuf &b) -> std::string& { return b.s; }, true, sam, sam_off);
}
extern "C" int ssg_sam_format(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, int n_pairs,
                              const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals, const char *const *comments,
                              const char *rg_id, char **sam, int64_t *sam_off)
{
	return ssg_sam_format_sel(idx, opt, res, 0, n_pairs, names, seq, off, quals, comments, rg_id, sam, sam_off);
}
/* single-end results (ssg_mem_process_reads): the lines of n_reads reads, sam_off[n_reads + 1] */
extern "C" int ssg_sam_format_se(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, int n_reads,
                                 const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals, const char *const *comments,
                                 const char *rg_id, char **sam, int64_t *sam_off)
{
	static thread_local std::vector<sbuf> keep;
	return format_all(idx, opt, res, 0, n_reads, names, seq, off, quals, comments, rg_id, keep, [](sbuf &b) -> std::string& { return b.s; }, true, sam, sam_off, true);
}
extern "C" int ssg_bam_format(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, int n_pairs,
                              const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals,
                              const char *rg_id, uint8_t **bam, int64_t *bam_off)
{
	static thread_local std::vector<bbuf> keep;
	return format_all(idx, opt, res, 0, n_pairs, names, seq, off, quals, 0, rg_id, keep, [](bbuf &b) -> std::vector<uint8_t>& { return b.b; }, false, (char**)bam, bam_off);
}
