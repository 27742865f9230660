/*
 * k_bam.h -- the two text ends of the hot path on the device.
 *
 * K1  FASTQ text -> reads: ssg_k_fq_unpack finds, for every record start the host's line scanner handed over (host/ranksplit.h: plain
 *     four-line records only), the name klib's kseq_read would return (/root/reference/src/samtools-1.3.1/htslib-1.3.1/htslib/kseq.h:189-229:
 *     '@', the name up to the first blank, the rest of the line as comment, one sequence line, a '+' line, one quality line) after upstream
 *     bseq.c's trim_readno, and where its bases and qualities lie; ssg_k_fq_codes turns the bases into nt4 codes (bwa's nst_nt4_table).
 * K11 alignment records -> BAM record bytes: what `sambamba view -S -f bam` (the next stage of the reference's pipeline,
 *     /root/reference/bin/speedseq:440) makes of upstream mem_aln2sam's line -- htslib sam_parse1's typing rules (sam.c:835-1028) in
 *     bam_write1's layout (sam.c:443-473), bin = hts_reg2bin (hts.h:580-586).  Field for field the host formatter of sam_format.cpp
 *     (aln2bam), which stays as the reference the tests compare this kernel with.
 * Both are byte movers: a lane per pair walks its own few hundred bytes; the only arithmetic is decimal printing.  HBM-bound by design
 * (about 0.7 KB read and 0.7 KB written per pair); DESIGN.md section 4 has the measured rates.
 */
#ifndef SSG_K_BAM_H
#define SSG_K_BAM_H
#include "ssg_dev.h"

/* where one read's text lies in the device copy of the input: name (without '@', trimmed), qualities (-1: none); the bases are in the code array */
typedef struct { int64_t name_off, qual_off; int32_t l_name, _pad; } ssg_rdtext_t;

/* ---------------- K1 ---------------- */
SSG_DEVFN int ssg_fq_blank(unsigned c) { return c == ' ' || (c - 9u) <= 4u; }   /* isspace in the C locale (kseq's delimiter 0) */

/* one lane per read: rec_off[r] = offset of the record's '@' in text[0 .. text_bytes).  len[r] = bases; err[0]: 1 = not a plain record; err[1] = the longest read */
__global__ void ssg_k_fq_unpack(long n_reads, const uint8_t *text, int64_t text_bytes, const int64_t *rec_off, ssg_rdtext_t *rd, int64_t *seq_at, int32_t *len, int32_t *err)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	int64_t p = rec_off[r];
	int bad = 0;
	if (p < 0 || p >= text_bytes || text[p] != '@') { bad = 1; p = 0; }
	const int64_t n0 = p + 1;
	int64_t q = n0;
	while (q < text_bytes && !ssg_fq_blank(text[q])) ++q;
	int l_name = (int)(q - n0);
	if (l_name > 2 && text[q - 2] == '/' && text[q - 1] >= '0' && text[q - 1] <= '9') l_name -= 2;   /* upstream trim_readno */
	while (q < text_bytes && text[q] != '\n') ++q;                                                  /* the comment, if any: not kept on this path */
	const int64_t s0 = q + 1;
	int64_t e = s0;
	while (e < text_bytes && text[e] != '\n') ++e;
	const int64_t L = e - s0;
	int64_t plus = e + 1, q0 = plus;
	if (plus >= text_bytes || text[plus] != '+') bad = 1;
	while (q0 < text_bytes && text[q0] != '\n') ++q0;
	++q0;
	if (q0 + L > text_bytes || (q0 + L < text_bytes && text[q0 + L] != '\n')) bad = 1;
	if (L <= 0 || L > 0x7fffffff || (L > 0 && text[s0 + L - 1] == '\r')) bad = 1;
	if (!bad) { const unsigned c = text[s0]; if (c == '@' || c == '+' || c == '>') bad = 1; }
	rd[r].name_off = n0; rd[r].l_name = l_name; rd[r].qual_off = q0; rd[r]._pad = 0;
	seq_at[r] = s0; len[r] = bad ? 0 : (int32_t)L;
	if (bad) atomicMax(err, 1);
	else if ((int32_t)L > err[1]) atomicMax(err + 1, (int32_t)L);   /* (a plain read of a stale maximum: at most a few atomics per distinct length) */
}

/* the two reads of a pair carry one name (upstream mem_sam_pe: "paired reads have different names"); err_pair = the first such pair + 1 */
__global__ void ssg_k_fq_pair_names(long n_pairs, const uint8_t *text, const ssg_rdtext_t *rd, int32_t *err_pair)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	const ssg_rdtext_t a = rd[2 * p], b = rd[2 * p + 1];
	int diff = a.l_name != b.l_name;
	for (int i = 0; !diff && i < a.l_name; ++i) diff = text[a.name_off + i] != text[b.name_off + i];
	if (diff) atomicMin(err_pair, (int32_t)(p + 1));
}

/* bases -> nt4 codes, a wave per read (coalesced on both sides) */
__global__ void ssg_k_fq_codes(long n_reads, const uint8_t *text, const int64_t *seq_at, const int64_t *off, uint8_t *seq)
{
	const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
	const int lane = wv_lane();
	for (long r = w0; r < n_reads; r += nw) {
		const uint8_t *s = text + seq_at[r]; uint8_t *d = seq + off[r];
		const int L = (int)(off[r + 1] - off[r]);
		for (int i = lane; i < L; i += 64) {
			const unsigned c = s[i] & 0xdfu;      /* bit 5 cleared: 'a' and 'A' alike, and nothing else becomes a letter */
			d[i] = (uint8_t)(c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4);
		}
	}
}

/* ---------------- K11 ---------------- */
typedef struct {
	const ssg_alnreq_t *req; const ssg_aln_t *alns; const int64_t *req_off;     /* the records of the batch (pe_core) */
	const uint8_t *seq; const int64_t *off;                                    /* nt4 codes */
	const uint8_t *text; const ssg_rdtext_t *rd;                               /* names and qualities */
	const char *ctg_names; const int32_t *ctg_name_off;                        /* contig i = ctg_names[ctg_name_off[i] .. ctg_name_off[i + 1]) */
	const char *rg_id; int32_t l_rg;                                           /* RG:Z value (l_rg = 0: no tag) */
	int32_t softclip;                                                          /* upstream MEM_F_SOFTCLIP (-Y) */
} ssg_bam_ctx_t;

/* two sinks with one interface: the size pass counts, the write pass stores */
struct ssg_bam_count_t {
	int64_t n;
	SSG_DEVMEM void b(unsigned) { ++n; }
	SSG_DEVMEM void w32(uint32_t) { n += 4; }
	SSG_DEVMEM void bytes(const uint8_t *, int k) { n += k; }
	SSG_DEVMEM void skip(int k) { n += k; }
	SSG_DEVMEM void set32(int64_t, uint32_t) {}
	SSG_DEVMEM void seq4(const uint8_t *, int, int, int) {}
	SSG_DEVMEM void qual(const uint8_t *, int, int, int) {}
	SSG_DEVMEM void fill(unsigned, int) {}
	static constexpr bool writes = false;
};
struct ssg_bam_store_t {
	uint8_t *base; int64_t n;
	SSG_DEVMEM void b(unsigned v) { base[n++] = (uint8_t)v; }
	SSG_DEVMEM void w32(uint32_t v) { base[n] = (uint8_t)v; base[n + 1] = (uint8_t)(v >> 8); base[n + 2] = (uint8_t)(v >> 16); base[n + 3] = (uint8_t)(v >> 24); n += 4; }
	SSG_DEVMEM void bytes(const uint8_t *s, int k) { for (int i = 0; i < k; ++i) base[n + i] = s[i]; n += k; }
	SSG_DEVMEM void skip(int k) { n += k; }
	SSG_DEVMEM void set32(int64_t at, uint32_t v) { base[at] = (uint8_t)v; base[at + 1] = (uint8_t)(v >> 8); base[at + 2] = (uint8_t)(v >> 16); base[at + 3] = (uint8_t)(v >> 24); }
	/* l bases from codes s[from], stepping dir (+1: as read; -1: complemented, from the end), two per byte, high nibble first; at n - ((l + 1) >> 1) .. */
	SSG_DEVMEM void seq4(const uint8_t *s, int from, int dir, int l)
	{
		uint8_t *d = base + n;
		for (int i = 0; i < l; i += 2) {
			const unsigned c0 = s[from + dir * i], c1 = i + 1 < l ? s[from + dir * (i + 1)] : 5u;
			const unsigned f0 = c0 > 3 ? 15u : dir > 0 ? 1u << c0 : 8u >> c0, f1 = c1 == 5u ? 0u : c1 > 3 ? 15u : dir > 0 ? 1u << c1 : 8u >> c1;
			d[i >> 1] = (uint8_t)(f0 << 4 | f1);
		}
	}
	SSG_DEVMEM void qual(const uint8_t *s, int from, int dir, int l) { uint8_t *d = base + n; for (int i = 0; i < l; ++i) d[i] = (uint8_t)(s[from + dir * i] - 33); }
	SSG_DEVMEM void fill(unsigned v, int l) { uint8_t *d = base + n; for (int i = 0; i < l; ++i) d[i] = (uint8_t)v; }
	static constexpr bool writes = true;
};

SSG_DEVFN int ssg_dec_len(int64_t v) { int k = v < 0 ? 2 : 1; uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v; while (u >= 10) { u /= 10; ++k; } return k; }
template <class S> SSG_DEVFN void ssg_put_dec(S &o, int64_t v)
{
	if (!S::writes) { o.skip(ssg_dec_len(v)); return; }
	char b[24]; int n = 24; uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
	do { b[--n] = (char)('0' + u % 10); u /= 10; } while (u);
	if (v < 0) b[--n] = '-';
	for (; n < 24; ++n) o.b((unsigned)b[n]);
}
/* sam_parse1's integer typing (sam.c:964-988): the smallest type that holds the value */
template <class S> SSG_DEVFN void ssg_bam_tag_int(S &o, char t0, char t1, int64_t v)
{
	o.b((unsigned)t0); o.b((unsigned)t1);
	if (v < 0) {
		if (v >= -128) { o.b('c'); o.b((unsigned)(v & 0xff)); }
		else if (v >= -32768) { o.b('s'); o.b((unsigned)(v & 0xff)); o.b((unsigned)(v >> 8 & 0xff)); }
		else { o.b('i'); o.w32((uint32_t)(int32_t)v); }
	} else {
		if (v <= 255) { o.b('C'); o.b((unsigned)v); }
		else if (v <= 65535) { o.b('S'); o.b((unsigned)(v & 0xff)); o.b((unsigned)(v >> 8)); }
		else { o.b('I'); o.w32((uint32_t)v); }
	}
}
SSG_DEVFN int ssg_bam_reg2bin(int64_t beg, int64_t end)
{	/* hts_reg2bin(beg, end, 14, 5), hts.h:580-586 */
	int l, s = 14, t = ((1 << 15) - 1) / 7;
	for (--end, l = 5; l > 0; --l, s += 3, t -= 1 << ((l << 1) + l)) if (beg >> s == end >> s) return t + (int)(beg >> s);
	return 0;
}
SSG_DEVFN int ssg_cigar_rlen(int n_cigar, const uint32_t *cigar)
{
	int l = 0;
	for (int k = 0; k < n_cigar; ++k) { const int op = (int)(cigar[k] & 0xf); if (op == 0 || op == 2) l += (int)(cigar[k] >> 4); }
	return l;
}
template <class S> SSG_DEVFN void ssg_put_ctg(S &o, const ssg_bam_ctx_t &c, int rid) { const int a = c.ctg_name_off[rid], e = c.ctg_name_off[rid + 1]; o.bytes((const uint8_t*)c.ctg_names + a, e - a); }

/* the record of main line `which` of read r (requests [g0, g1): main lines first among the kinds, XA entries with their owner); mate = the
 * first main line of the other read, as mem_sam_pe hands it to mem_aln2sam */
template <class S> SSG_DEVFN void ssg_bam_record(S &o, const ssg_bam_ctx_t &c, long r, int64_t g0, int64_t g1, int64_t gw, int which, const ssg_aln_t *mate)
{
	const ssg_aln_t &a = c.alns[gw];
	int flag = a.flag, rid = a.rid, is_rev = a.is_rev, n_cigar = a.n_cigar; int64_t pos = a.pos;
	int m_rid = mate->rid, m_is_rev = mate->is_rev, m_n_cigar = mate->n_cigar; int64_t m_pos = mate->pos;
	flag |= 0x1;
	flag |= rid < 0 ? 0x4 : 0;
	flag |= m_rid < 0 ? 0x8 : 0;
	if (rid < 0 && m_rid >= 0) { rid = m_rid; pos = m_pos; is_rev = m_is_rev; n_cigar = 0; }
	if (m_rid < 0 && rid >= 0) { m_rid = rid; m_pos = pos; m_is_rev = is_rev; m_n_cigar = 0; }
	flag |= is_rev ? 0x10 : 0;
	flag |= m_is_rev ? 0x20 : 0;
	const int64_t base = o.n;
	o.skip(36);
	const ssg_rdtext_t t = c.rd[r];
	o.bytes(c.text + t.name_off, t.l_name); o.b(0);
	const int l_qname = t.l_name + 1;
	int32_t tid = -1, bpos = -1, mapq = 0; int64_t rl = 0;
	if (rid >= 0) {
		tid = rid; bpos = (int32_t)pos; mapq = a.mapq;
		for (int i = 0; i < n_cigar; ++i) {
			int op = (int)(a.cigar[i] & 0xf);
			if (!c.softclip && (op == 3 || op == 4)) op = which ? 4 : 3;
			const uint32_t len = a.cigar[i] >> 4;
			o.w32(len << 4 | (uint32_t)(op <= 2 ? op : op + 1));   /* "MIDSH" -> BAM codes M0 I1 D2 S4 H5 */
			if (op == 0 || op == 2) rl += len;
		}
		if (!n_cigar) flag |= 4;            /* sam_parse1: a record without CIGAR is treated as unmapped */
	} else { n_cigar = 0; flag |= 4; }
	const int64_t rlen = (!(flag & 4) && n_cigar) ? rl : 1;
	const int bin = ssg_bam_reg2bin(bpos, bpos + rlen);
	int32_t mtid = -1, mpos = -1, isize = 0;
	if (m_rid >= 0) {
		mtid = m_rid; mpos = (int32_t)m_pos;
		if (rid == m_rid) {
			const int64_t p0 = pos + (is_rev ? ssg_cigar_rlen(n_cigar, a.cigar) - 1 : 0);
			const int64_t p1 = m_pos + (m_is_rev ? ssg_cigar_rlen(m_n_cigar, mate->cigar) - 1 : 0);
			if (!(m_n_cigar == 0 || n_cigar == 0)) isize = (int32_t)(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		}
	}
	int32_t l_qseq = 0;
	if (!(flag & 0x100)) {
		const int l_seq = (int)(c.off[r + 1] - c.off[r]);
		int qb = 0, qe = l_seq;
		const bool cl0 = n_cigar && which && !c.softclip && ((a.cigar[0] & 0xf) == 4 || (a.cigar[0] & 0xf) == 3);
		const bool cl1 = n_cigar && which && !c.softclip && ((a.cigar[n_cigar - 1] & 0xf) == 4 || (a.cigar[n_cigar - 1] & 0xf) == 3);
		if (!is_rev) { if (cl0) qb += (int)(a.cigar[0] >> 4); if (cl1) qe -= (int)(a.cigar[n_cigar - 1] >> 4); }
		else { if (cl0) qe -= (int)(a.cigar[0] >> 4); if (cl1) qb += (int)(a.cigar[n_cigar - 1] >> 4); }
		l_qseq = qe > qb ? qe - qb : 0;
		const uint8_t *s = c.seq + c.off[r];
		o.seq4(s, is_rev ? qe - 1 : qb, is_rev ? -1 : 1, l_qseq); o.skip((l_qseq + 1) >> 1);
		if (t.qual_off < 0) o.fill(0xff, l_qseq); else o.qual(c.text + t.qual_off, is_rev ? qe - 1 : qb, is_rev ? -1 : 1, l_qseq);
		o.skip(l_qseq);
	}
	if (n_cigar) {
		ssg_bam_tag_int(o, 'N', 'M', a.NM);
		o.b('M'); o.b('D'); o.b('Z'); o.bytes((const uint8_t*)a.md, a.l_md); o.b(0);
	}
	if (a.score >= 0) ssg_bam_tag_int(o, 'A', 'S', a.score);
	if (a.sub >= 0) ssg_bam_tag_int(o, 'X', 'S', a.sub);
	if (c.l_rg > 0) { o.b('R'); o.b('G'); o.b('Z'); o.bytes((const uint8_t*)c.rg_id, c.l_rg); o.b(0); }
	if (!(flag & 0x100)) {
		bool any = false; int k = 0;
		for (int64_t g = g0; g < g1; ++g) { if (c.req[g].kind != SSG_REQ_MAIN) continue; if (k != which && !(c.alns[g].flag & 0x100)) any = true; ++k; }
		if (any) {
			o.b('S'); o.b('A'); o.b('Z');
			k = 0;
			for (int64_t g = g0; g < g1; ++g) {
				if (c.req[g].kind != SSG_REQ_MAIN) continue;
				const int me = k++;
				const ssg_aln_t &x = c.alns[g];
				if (me == which || (x.flag & 0x100)) continue;
				ssg_put_ctg(o, c, x.rid); o.b(',');
				ssg_put_dec(o, x.pos + 1); o.b(',');
				o.b(x.is_rev ? '-' : '+'); o.b(',');
				for (int j = 0; j < x.n_cigar; ++j) { ssg_put_dec(o, (int64_t)(x.cigar[j] >> 4)); o.b((unsigned)"MIDSH"[x.cigar[j] & 0xf]); }
				o.b(','); ssg_put_dec(o, x.mapq);
				o.b(','); ssg_put_dec(o, x.NM);
				o.b(';');
			}
			o.b(0);
		}
	}
	if (a.rid >= 0) {   /* XA: the entries of this line's region, in request order (upstream mem_gen_alt's string) */
		const int owner = c.req[gw].owner;
		bool open = false;
		for (int64_t g = g0; g < g1; ++g) {
			if (c.req[g].kind != SSG_REQ_XA || c.req[g].owner != owner) continue;
			const ssg_aln_t &x = c.alns[g];
			if (!open) { o.b('X'); o.b('A'); o.b('Z'); open = true; }
			ssg_put_ctg(o, c, x.rid); o.b(','); o.b(x.is_rev ? '-' : '+'); ssg_put_dec(o, x.pos + 1); o.b(',');
			for (int j = 0; j < x.n_cigar; ++j) { ssg_put_dec(o, (int64_t)(x.cigar[j] >> 4)); o.b((unsigned)"MIDSHN"[x.cigar[j] & 0xf]); }
			o.b(','); ssg_put_dec(o, x.NM); o.b(';');
		}
		if (open) o.b(0);
	}
	if (S::writes) {
		o.set32(base, (uint32_t)(o.n - base - 4));
		o.set32(base + 4, (uint32_t)tid); o.set32(base + 8, (uint32_t)bpos);
		o.set32(base + 12, (uint32_t)bin << 16 | (uint32_t)(mapq & 0xff) << 8 | (uint32_t)l_qname);
		o.set32(base + 16, (uint32_t)((flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0)) << 16 | (uint32_t)n_cigar);
		o.set32(base + 20, (uint32_t)l_qseq); o.set32(base + 24, (uint32_t)mtid); o.set32(base + 28, (uint32_t)mpos); o.set32(base + 32, (uint32_t)isize);
	}
}

/* all records of pair p through sink o; returns the number of records; *cand: the pair can reach one of samblaster's side streams under ANY of its
 * options (a read with several main lines: splitter test; both ends mapped without the proper-pair flag: discordant test) */
template <class S> SSG_DEVFN int ssg_bam_pair(S &o, const ssg_bam_ctx_t &c, long p, int *cand)
{
	int n_rec = 0, nmain[2];
	const ssg_aln_t *first[2];
	for (int i = 0; i < 2; ++i) {
		const long r = 2 * p + i;
		const int64_t g0 = c.req_off[r], g1 = c.req_off[r + 1];
		first[i] = 0; nmain[i] = 0;
		for (int64_t g = g0; g < g1; ++g) if (c.req[g].kind == SSG_REQ_MAIN) { if (!first[i]) first[i] = c.alns + g; ++nmain[i]; }
	}
	if (!first[0] || !first[1]) { *cand = -1; return 0; }   /* every read has at least one main line (an unmapped record); reported by the caller */
	for (int i = 0; i < 2; ++i) {
		const long r = 2 * p + i;
		const int64_t g0 = c.req_off[r], g1 = c.req_off[r + 1];
		int which = 0;
		for (int64_t g = g0; g < g1; ++g) {
			if (c.req[g].kind != SSG_REQ_MAIN) continue;
			ssg_bam_record(o, c, r, g0, g1, g, which, first[!i]);
			++which; ++n_rec;
		}
	}
	const ssg_aln_t &a1 = c.alns[c.req_off[2 * p]], &a2 = c.alns[c.req_off[2 * p + 1]];
	*cand = nmain[0] > 1 || nmain[1] > 1 || (a1.rid >= 0 && a2.rid >= 0 && !(a1.flag & 0x2));
	return n_rec;
}

__global__ void __launch_bounds__(64) ssg_k_bam_size(ssg_bam_ctx_t c, long n_pairs, int32_t *bytes, int32_t *n_rec, int32_t *cand, int32_t *err)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	ssg_bam_count_t o; o.n = 0;
	int cd = 0;
	const int nr = ssg_bam_pair(o, c, p, &cd);
	if (cd < 0 || o.n > 0x7fffffff) { atomicMax(err, 1); cd = 0; }
	bytes[p] = (int32_t)o.n; n_rec[p] = nr; cand[p] = cd;
}

__global__ void __launch_bounds__(64) ssg_k_bam_write(ssg_bam_ctx_t c, long n_pairs, const int64_t *byte_off, const int64_t *rec_off, const int64_t *cand_off, const int32_t *cand,
                                                     uint8_t *out, ssg_bam_cand_t *cands, int32_t *err)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	ssg_bam_store_t o; o.base = out + byte_off[p]; o.n = 0;
	int cd = 0;
	const int nr = ssg_bam_pair(o, c, p, &cd);
	if (o.n != byte_off[p + 1] - byte_off[p] || nr != (int)(rec_off[p + 1] - rec_off[p])) atomicMax(err, 2);   /* the two passes disagree: never seen; fails the call */
	if (cand[p]) { ssg_bam_cand_t x; x.pair = p; x.first_rec = rec_off[p]; x.n_rec = nr; x.byte_off = byte_off[p]; x.n_bytes = o.n; cands[cand_off[p]] = x; }
}
#endif
