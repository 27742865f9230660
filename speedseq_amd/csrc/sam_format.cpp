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
#include "../../include/ssgpu.h"

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
             int n, const ssg_aln_t *const *list, int which, const mate_t *m_, const std::string *XA, const char *rg_id)
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
				if (c == 3 || c == 4) c = which ? 4 : 3;
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
		if (n_cigar && which) {
			if ((a.cigar[0] & 0xf) == 4 || (a.cigar[0] & 0xf) == 3) qb += a.cigar[0] >> 4;
			if ((a.cigar[n_cigar-1] & 0xf) == 4 || (a.cigar[n_cigar-1] & 0xf) == 3) qe -= a.cigar[n_cigar-1] >> 4;
		}
		{ char *d = str.grow((size_t)(qe > qb ? qe - qb : 0)); for (int i = qb; i < qe; ++i) *d++ = "ACGTN"[seq[i]]; }
		str.putc('\t');
		if (qual) str.s.append(qual + qb, qe - qb); else str.putc('*');
	} else {
		int qb = 0, qe = l_seq;
		if (n_cigar && which) {
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
} // namespace

/* pairs [p0, p1) into out; sam_off entries relative to out's start */
static int format_range(const ssg_index_t *idx, const ssg_pe_result_t *res, int p0, int p1, const char *const *names, const uint8_t *seq, const int64_t *off,
                        const char *const *quals, const char *const *comments, const char *rg_id, sbuf &out, int64_t *sam_off)
{
	const int64_t *req_off = ssg_pe_req_off(res);
	const ssg_alnreq_t *req = ssg_pe_req(res);
	const ssg_aln_t *alns = ssg_pe_alns(res);
	std::vector<const ssg_aln_t*> mains[2];
	std::vector<std::string> xa[2];
	std::vector<int> owner;
	for (int p = p0; p < p1; ++p) {
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
			sam_off[r] = (int64_t)out.s.size();
			const int l_seq = (int)(off[r+1] - off[r]);
			for (size_t k = 0; k < mains[i].size(); ++k)
				aln2sam(idx, out, names[r], l_seq, seq + off[r], quals ? quals[r] : 0, comments ? comments[r] : 0,
				        (int)mains[i].size(), mains[i].data(), (int)k, &mate[!i], &xa[i][k], rg_id);
		}
	}
	return 0;
}

/* Text assembly is split over host threads by ranges of pairs (opt->n_threads, as upstream's worker2 threads print), then joined in input order. */
extern "C" int ssg_sam_format(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, int n_pairs,
                              const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals, const char *const *comments,
                              const char *rg_id, char **sam, int64_t *sam_off)
{
	int T = opt && opt->n_threads > 1 ? opt->n_threads : 1;
	{ const char *e = getenv("SSG_FMT_THREADS"); if (e && atoi(e) > 0) T = atoi(e); }
	if (T > 64) T = 64;
	if (T > (n_pairs + 1023) / 1024) T = (n_pairs + 1023) / 1024;
	if (T < 1) T = 1;
	/* the per-thread text buffers persist across calls of the calling thread (bwa's formatter thread): after the first batch they
	 * are warm memory instead of hundreds of MB of fresh pages per call */
	static thread_local std::vector<sbuf> outs_keep;
	if ((int)outs_keep.size() < T) outs_keep.resize(T);
	std::vector<sbuf> &outs = outs_keep;
	std::vector<int> rcs(T, 0); std::vector<std::thread> th;
	auto lo = [&](int t) { return (int)((int64_t)n_pairs * t / T); };
	for (int t = 0; t < T; ++t) {
		auto work = [&, t]() { outs[t].s.clear(); outs[t].s.reserve((size_t)(lo(t + 1) - lo(t)) * 1000); rcs[t] = format_range(idx, res, lo(t), lo(t + 1), names, seq, off, quals, comments, rg_id, outs[t], sam_off); };
		if (T == 1) work(); else th.emplace_back(work);
	}
	for (auto &x : th) x.join();
	th.clear();
	size_t tot = 0; std::vector<size_t> base(T + 1, 0);
	for (int t = 0; t < T; ++t) { if (rcs[t]) return rcs[t]; base[t] = tot; tot += outs[t].s.size(); }
	char *buf = (char*)malloc(tot + 1);
	if (!buf) return SSG_ENOMEM;
	for (int t = 0; t < T; ++t) {   /* joined in input order, each part copied by its own thread */
		auto work = [&, t]() { memcpy(buf + base[t], outs[t].s.data(), outs[t].s.size()); for (int r = 2 * lo(t); r < 2 * lo(t + 1); ++r) sam_off[r] += (int64_t)base[t]; };
		if (T == 1) work(); else th.emplace_back(work);
	}
	for (auto &x : th) x.join();
	buf[tot] = 0; sam_off[2 * n_pairs] = (int64_t)tot;
	*sam = buf;
	return 0;
}
