/*
 * bamio.h -- BAM / BGZF / BAI as the reference's pipeline reads and writes them.  Formats follow the
 * in-tree htslib 1.3.1:
 *   BGZF block        /root/reference/src/samtools-1.3.1/htslib-1.3.1/bgzf.c:298-342 (18-byte header with the BC subfield,
 *                     raw deflate, CRC32 + ISIZE; payload <= 0xff00, bgzf.h:43; 28-byte EOF block)
 *   BAM record        sam.c:443-473 (bam_write1: block_size, refID, pos, bin<<16|mapq<<8|l_qname, flag<<16|n_cigar, l_seq,
 *                     mate refID / pos, tlen, qname, cigar, 4-bit seq, qual, aux)
 *   SAM text -> BAM   sam.c:835-1028 (sam_parse1: field rules, smallest-integer aux types, '*' conventions, bin from
 *                     hts.h:580-586 reg2bin)
 *   coordinate order  bam_sort.c:1607-1614 (key tid<<32 | (pos+1)<<1 | reverse; stable)
 *   BAI               hts.c:1192-1495 (hts_idx_push / finish / compress_binning / save; linear index 16 kb windows,
 *                     pseudo-bin 37450 with file range and mapped / unmapped counts)
 * (De)compression of blocks runs on host threads (zlib); the coordinate sort of the keys runs on the MI355X through libssgpu.
 */
#ifndef SSG_BAMIO_H
#define SSG_BAMIO_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <errno.h>
#include <unistd.h>
#include <zlib.h>
#include <string>
#include <vector>
#include <map>
#include <unordered_map>
#include <thread>
#include <functional>
#include <algorithm>

#define BGZF_MAX_PAYLOAD 0xff00
static const uint8_t BGZF_EOF[28] = { 0x1f,0x8b,0x08,0x04,0,0,0,0,0,0xff,0x06,0,0x42,0x43,0x02,0,0x1b,0,0x03,0,0,0,0,0,0,0,0,0 };

static inline void parallel_for(int n_threads, size_t n, const std::function<void(size_t, size_t, int)> &fn)
{	/* fn(begin, end, thread) over contiguous ranges */
	if (n_threads < 1) n_threads = 1;
	if ((size_t)n_threads > n) n_threads = n ? (int)n : 1;
	if (n_threads == 1) { fn(0, n, 0); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t) th.emplace_back([&, t]() { fn(n * t / n_threads, n * (t + 1) / n_threads, t); });
	for (auto &x : th) x.join();
}

static inline void io_write_all(int fd, const void *p, size_t n)
{
	const char *c = (const char*)p;
	while (n) { ssize_t w = write(fd, c, n); if (w < 0) { if (errno == EINTR) continue; perror("write"); exit(1); } c += w; n -= (size_t)w; }
}

/* one BGZF block from <= 0xff00 payload bytes; level 0 = stored; returns the block size.
 * The deflate state is the calling thread's own and lives as long as the thread: deflateInit2 allocates and clears ~256 KB, and a
 * sort writes tens of thousands of blocks from hundreds of threads -- that many allocations and releases per second meet in the
 * allocator and in the kernel's memory-map lock (round 3's sort wrote 0.8 GB/s of records on 256 threads); deflateReset keeps the
 * memory and only clears the hash heads. */
struct bgzf_deflater_t {
	z_stream zs; int level; bool live;
	bgzf_deflater_t() : level(-2), live(false) { memset(&zs, 0, sizeof(zs)); }
	~bgzf_deflater_t() { if (live) deflateEnd(&zs); }
	bool ready(int lvl)
	{
		if (live && lvl == level) return deflateReset(&zs) == Z_OK;
		if (live) { deflateEnd(&zs); live = false; }
		memset(&zs, 0, sizeof(zs));
		if (deflateInit2(&zs, lvl, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
		live = true; level = lvl; return true;
	}
};
static inline size_t bgzf_make_block(const uint8_t *src, size_t slen, int level, uint8_t *dst /* >= 65536 */)
{
	static const uint8_t hdr[16] = { 0x1f,0x8b,0x08,0x04,0,0,0,0,0,0xff,0x06,0,0x42,0x43,0x02,0 };
	memcpy(dst, hdr, 16);
	size_t clen;
	if (level == 0) {
		uint8_t *d = dst + 18;
		d[0] = 1; d[1] = (uint8_t)(slen & 0xff); d[2] = (uint8_t)(slen >> 8); d[3] = (uint8_t)~d[1]; d[4] = (uint8_t)~d[2];
		memcpy(d + 5, src, slen); clen = slen + 5;
	} else {
		static thread_local bgzf_deflater_t df;
		if (!df.ready(level)) { fprintf(stderr, "[sambamba] deflateInit2 failed\n"); exit(1); }
		z_stream &zs = df.zs;
		zs.next_in = (Bytef*)src; zs.avail_in = (uInt)slen; zs.next_out = dst + 18; zs.avail_out = 65536 - 18 - 8;
		if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return bgzf_make_block(src, slen, 0, dst);   /* incompressible payload: store it instead */
		clen = zs.total_out;
	}
	const size_t bsize = 18 + clen + 8;
	dst[16] = (uint8_t)((bsize - 1) & 0xff); dst[17] = (uint8_t)((bsize - 1) >> 8);
	const uint32_t crc = (uint32_t)crc32(crc32(0L, 0, 0), src, (uInt)slen), isz = (uint32_t)slen;
	memcpy(dst + 18 + clen, &crc, 4); memcpy(dst + 18 + clen + 4, &isz, 4);
	return bsize;
}

/* BGZF writer: payload accumulates in blocks (a record is not split across blocks unless it is larger than one, as bam_write1's
 * bgzf_flush_try arranges); full blocks are compressed by the thread pool in batches and written in order */
struct bgzf_out_t {
	int fd, level, threads; std::vector<uint8_t> data; std::vector<size_t> cut;   /* cut: payload end of every closed block; the open block is data[cut.back()..] */
	bgzf_out_t(int fd_, int level_, int threads_) : fd(fd_), level(level_ < 0 ? 6 : level_), threads(threads_) { cut.push_back(0); data.reserve((size_t)64 << 20); }
	size_t open_len() const { return data.size() - cut.back(); }
	void close_block() { if (open_len()) cut.push_back(data.size()); }
	void put(const void *p, size_t n)
	{
		const uint8_t *c = (const uint8_t*)p;
		while (n) {
			size_t room = BGZF_MAX_PAYLOAD - open_len();
			if (!room) { close_block(); room = BGZF_MAX_PAYLOAD; }
			const size_t k = n < room ? n : room;
			data.insert(data.end(), c, c + k); c += k; n -= k;
		}
		if (open_len() == BGZF_MAX_PAYLOAD) close_block();
		if (cut.size() > 1024) drain(false);
	}
	void record(const void *p, size_t n) { if (open_len() + n > BGZF_MAX_PAYLOAD) close_block(); put(p, n); }   /* bgzf_flush_try + bgzf_write */
	void drain(bool all)
	{
		if (all) close_block();
		const size_t nb = cut.size() - 1;
		if (!nb) return;
		std::vector<std::vector<uint8_t> > outb(nb);
		parallel_for(threads, nb, [&](size_t b0, size_t b1, int) {
			for (size_t b = b0; b < b1; ++b) { outb[b].resize(65536); outb[b].resize(bgzf_make_block(data.data() + cut[b], cut[b + 1] - cut[b], level, outb[b].data())); }
		});
		for (size_t b = 0; b < nb; ++b) io_write_all(fd, outb[b].data(), outb[b].size());
		const size_t done = cut[nb];
		data.erase(data.begin(), data.begin() + done);
		cut.assign(1, 0);
	}
	void finish() { drain(true); io_write_all(fd, BGZF_EOF, 28); }
};

/* BGZF reader: blocks are read sequentially and inflated by the pool in batches; exposes the payload as a byte stream and, for
 * the indexer, the virtual file offset bgzf_tell would report */
struct bgzf_in_t {
	int fd, threads; bool eof; uint64_t caddr;                 /* compressed offset of the next block to read */
	std::vector<uint8_t> raw; size_t raw_pos;                  /* read-ahead of compressed bytes */
	struct blk_t { uint64_t addr; std::vector<uint8_t> data; };
	std::vector<blk_t> q; size_t qi, qo;                       /* current batch, block index, offset inside the block */
	bgzf_in_t(int fd_, int threads_) : fd(fd_), threads(threads_), eof(false), caddr(0), raw_pos(0), qi(0), qo(0) {}
	bool fill_raw(size_t need)
	{
		while (raw.size() - raw_pos < need && !eof) {
			const size_t old = raw.size(); raw.resize(old + ((size_t)8 << 20));
			ssize_t r = read(fd, raw.data() + old, (size_t)8 << 20);
			if (r < 0) { if (errno == EINTR) { raw.resize(old); continue; } perror("[sambamba] read"); exit(1); }
			raw.resize(old + (size_t)r);
			if (r == 0) eof = true;
		}
		return raw.size() - raw_pos >= need;
	}
	bool next_batch()
	{	/* up to 512 blocks */
		struct span_t { size_t off, len; uint64_t addr; };
		std::vector<span_t> sp;
		if (raw_pos > ((size_t)8 << 20)) { raw.erase(raw.begin(), raw.begin() + raw_pos); raw_pos = 0; }   /* consumed bytes go here, between batches: the spans below are offsets into raw */
		while (sp.size() < 512) {
			if (!fill_raw(18)) break;
			const uint8_t *h = raw.data() + raw_pos;
			if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { fprintf(stderr, "[sambamba] not a BGZF block\n"); exit(1); }
			const size_t xlen = h[10] | (size_t)h[11] << 8;
			if (!fill_raw(12 + xlen)) { fprintf(stderr, "[sambamba] truncated BGZF header\n"); exit(1); }
			h = raw.data() + raw_pos;
			size_t bsize = 0;
			for (size_t o = 12; o + 4 <= 12 + xlen; ) { const size_t sl = h[o + 2] | (size_t)h[o + 3] << 8; if (h[o] == 'B' && h[o + 1] == 'C' && sl == 2) bsize = (size_t)(h[o + 4] | (size_t)h[o + 5] << 8) + 1; o += 4 + sl; }
			if (!bsize || !fill_raw(bsize)) { fprintf(stderr, "[sambamba] truncated BGZF block\n"); exit(1); }
			span_t s; s.off = raw_pos; s.len = bsize; s.addr = caddr; sp.push_back(s);
			raw_pos += bsize; caddr += bsize;
		}
		if (sp.empty()) return false;
		q.assign(sp.size(), blk_t());
		parallel_for(threads, sp.size(), [&](size_t b0, size_t b1, int) {
			for (size_t b = b0; b < b1; ++b) {
				const uint8_t *h = raw.data() + sp[b].off; const size_t xlen = h[10] | (size_t)h[11] << 8;
				uint32_t isz; memcpy(&isz, h + sp[b].len - 4, 4);
				q[b].addr = sp[b].addr; q[b].data.resize(isz);
				const uint8_t *df = h + 12 + xlen; const size_t dn = sp[b].len - 12 - xlen - 8;
				if (isz && dn == (size_t)isz + 5 && df[0] == 1 && (df[1] | (size_t)df[2] << 8) == isz && ((df[1] | (size_t)df[2] << 8) ^ 0xffffu) == (df[3] | (size_t)df[4] << 8)) {
					memcpy(q[b].data.data(), df + 5, isz);          /* one final stored block (`view -l 0`, the shape the reference's pipeline feeds the sort): no inflate state to set up */
				} else if (isz) {
					z_stream zs; memset(&zs, 0, sizeof(zs));
					zs.next_in = (Bytef*)(h + 12 + xlen); zs.avail_in = (uInt)(sp[b].len - 12 - xlen - 8); zs.next_out = q[b].data.data(); zs.avail_out = isz;
					if (inflateInit2(&zs, -15) != Z_OK || inflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "[sambamba] inflate failed\n"); exit(1); }
					inflateEnd(&zs);
				}
			}
		});
		qi = 0; qo = 0;
		return true;
	}
	/* bgzf_read: up to n bytes; returns the count (0 at end of file) */
	size_t get(void *dst, size_t n)
	{
		uint8_t *d = (uint8_t*)dst; size_t got = 0;
		while (got < n) {
			while (qi < q.size() && qo >= q[qi].data.size()) { ++qi; qo = 0; }
			if (qi >= q.size()) { if (!next_batch()) break; continue; }
			const size_t k = std::min(n - got, q[qi].data.size() - qo);
			memcpy(d + got, q[qi].data.data() + qo, k); got += k; qo += k;
		}
		return got;
	}
	/* advance n bytes without copying them; returns the count */
	size_t skip(size_t n)
	{
		size_t got = 0;
		while (got < n) {
			while (qi < q.size() && qo >= q[qi].data.size()) { ++qi; qo = 0; }
			if (qi >= q.size()) { if (!next_batch()) break; continue; }
			const size_t k = std::min(n - got, q[qi].data.size() - qo);
			got += k; qo += k;
		}
		return got;
	}
	/* bgzf_tell after the last get(): the block holding the next unread byte (an exhausted block reports the next block's address, offset 0) */
	uint64_t tell()
	{
		while (qi < q.size() && qo >= q[qi].data.size()) { ++qi; qo = 0; }
		if (qi >= q.size()) { if (!next_batch()) return caddr << 16; while (qi < q.size() && q[qi].data.empty()) ++qi; if (qi >= q.size()) return caddr << 16; }
		return q[qi].addr << 16 | (uint64_t)qo;
	}
};

/* ---- header ---- */
struct bam_hdr_t { std::string text; std::vector<std::string> names; std::vector<int32_t> lens; std::unordered_map<std::string, int> id; };
static inline void hdr_from_text(bam_hdr_t &h)
{	/* @SQ SN / LN in order (sam_hdr_parse) */
	h.names.clear(); h.lens.clear(); h.id.clear();
	size_t p = 0;
	while (p < h.text.size()) {
		size_t e = h.text.find('\n', p); if (e == std::string::npos) e = h.text.size();
		if (e - p > 3 && !h.text.compare(p, 3, "@SQ")) {
			std::string sn; long ln = 0; size_t q = p + 3;
			while (q < e) {
				size_t t = h.text.find('\t', q + 1); if (t == std::string::npos || t > e) t = e;
				if (t - q > 4 && !h.text.compare(q, 4, "\tSN:")) sn = h.text.substr(q + 4, t - q - 4);
				else if (t - q > 4 && !h.text.compare(q, 4, "\tLN:")) ln = atol(h.text.c_str() + q + 4);
				q = t;
			}
			h.id.emplace(sn, (int)h.names.size()); h.names.push_back(sn); h.lens.push_back((int32_t)ln);
		}
		p = e + 1;
	}
}
static inline void hdr_write(bgzf_out_t &o, const bam_hdr_t &h)
{	/* bam_hdr_write */
	std::vector<uint8_t> b;
	auto p32 = [&](int32_t v) { b.insert(b.end(), (uint8_t*)&v, (uint8_t*)&v + 4); };
	b.insert(b.end(), { 'B', 'A', 'M', 1 });
	p32((int32_t)h.text.size()); b.insert(b.end(), h.text.begin(), h.text.end());
	p32((int32_t)h.names.size());
	for (size_t i = 0; i < h.names.size(); ++i) { p32((int32_t)h.names[i].size() + 1); b.insert(b.end(), h.names[i].begin(), h.names[i].end()); b.push_back(0); p32(h.lens[i]); }
	o.put(b.data(), b.size());
	o.close_block();       /* bam_hdr_write ends with bgzf_flush: the records start a fresh block */
}
static inline bool hdr_read(bgzf_in_t &in, bam_hdr_t &h)
{
	char magic[4]; int32_t l_text, n_ref;
	if (in.get(magic, 4) != 4 || memcmp(magic, "BAM\1", 4)) return false;
	if (in.get(&l_text, 4) != 4) return false;
	h.text.resize((size_t)l_text);
	if (l_text && in.get(&h.text[0], (size_t)l_text) != (size_t)l_text) return false;
	while (!h.text.empty() && h.text.back() == 0) h.text.pop_back();
	if (in.get(&n_ref, 4) != 4) return false;
	h.names.clear(); h.lens.clear(); h.id.clear();
	for (int i = 0; i < n_ref; ++i) {
		int32_t l_name, l_ref;
		if (in.get(&l_name, 4) != 4) return false;
		std::string nm((size_t)l_name, 0);
		if (in.get(&nm[0], (size_t)l_name) != (size_t)l_name || in.get(&l_ref, 4) != 4) return false;
		nm.resize(strlen(nm.c_str()));
		h.id.emplace(nm, i); h.names.push_back(nm); h.lens.push_back(l_ref);
	}
	return true;
}

/* ---- SAM line -> BAM record (sam_parse1 + bam_write1's 36 leading bytes) ---- */
static inline int reg2bin(int64_t beg, int64_t end)
{	/* hts_reg2bin(beg, end, 14, 5) */
	int l, s = 14, t = ((1 << 15) - 1) / 7;
	for (--end, l = 5; l > 0; --l, s += 3, t -= 1 << ((l << 1) + l)) if (beg >> s == end >> s) return t + (int)(beg >> s);
	return 0;
}
static const char SEQ_NT16[] = "=ACMGRSVTWYHKDBN";
struct nt16_tab_t { uint8_t t[256]; nt16_tab_t() { memset(t, 15, 256); for (int i = 0; i < 16; ++i) { t[(uint8_t)SEQ_NT16[i]] = (uint8_t)i; t[(uint8_t)tolower(SEQ_NT16[i])] = (uint8_t)i; } } };
static inline const uint8_t *nt16_tab() { static nt16_tab_t T; return T.t; }

/* appends block_size + record to out; returns 0, or -1 on a malformed line (message in err) */
static inline int sam_line_to_bam(const char *s, const char *e, const bam_hdr_t &h, std::vector<uint8_t> &out, std::string &err)
{
	const char *f[11], *fe[11]; const char *p = s; int nf = 0;
	while (nf < 11) { const char *t = (const char*)memchr(p, '\t', (size_t)(e - p)); f[nf] = p; fe[nf] = t ? t : e; ++nf; if (!t) { p = e; break; } p = t + 1; }
	if (nf < 11) { err = "fewer than 11 fields"; return -1; }
	const char *aux = fe[10] < e ? fe[10] + 1 : e;
	const size_t base = out.size();
	out.resize(base + 36);
	auto tok = [&](int i) { return std::string(f[i], (size_t)(fe[i] - f[i])); };
	const size_t l_qname = (size_t)(fe[0] - f[0]) + 1;
	if (l_qname > 255) { err = "query name too long"; return -1; }
	out.insert(out.end(), f[0], fe[0]); out.push_back(0);
	int32_t flag = (int32_t)strtol(tok(1).c_str(), 0, 0);
	int32_t tid = -1;
	if (!(fe[2] - f[2] == 1 && *f[2] == '*')) { auto it = h.id.find(tok(2)); tid = it == h.id.end() ? -1 : it->second; }
	int32_t pos = (int32_t)strtol(tok(3).c_str(), 0, 10) - 1;
	if (pos < 0 && tid >= 0) tid = -1;
	if (tid < 0) flag |= 4;
	const int32_t mapq = (int32_t)strtol(tok(4).c_str(), 0, 10);
	int32_t n_cigar = 0; int64_t rlen = 1, qlen_c = 0;
	if (!(*f[5] == '*')) {
		for (const char *c = f[5]; c < fe[5]; ++c) if (!isdigit((unsigned char)*c)) ++n_cigar;
		if (n_cigar == 0) { err = "no CIGAR operations"; return -1; }
		if (n_cigar >= 65536) { err = "too many CIGAR operations"; return -1; }
		int64_t rl = 0; const char *c = f[5];
		for (int i = 0; i < n_cigar; ++i) {
			uint32_t len = 0; while (c < fe[5] && isdigit((unsigned char)*c)) len = len * 10 + (uint32_t)(*c++ - '0');
			const char *ops = "MIDNSHP=XB", *q = c < fe[5] ? strchr(ops, *c) : 0;
			if (!q || !*c) { err = "unrecognized CIGAR operator"; return -1; }
			const uint32_t op = (uint32_t)(q - ops), v = len << 4 | op; ++c;
			out.insert(out.end(), (uint8_t*)&v, (uint8_t*)&v + 4);
			if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += len;
			if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen_c += len;
		}
		rlen = !(flag & 4) ? rl : 1;
	} else flag |= 4;
	const int bin = reg2bin(pos, pos + rlen);
	int32_t mtid;
	if (fe[6] - f[6] == 1 && *f[6] == '=') mtid = tid;
	else if (fe[6] - f[6] == 1 && *f[6] == '*') mtid = -1;
	else { auto it = h.id.find(tok(6)); mtid = it == h.id.end() ? -1 : it->second; }
	int32_t mpos = (int32_t)strtol(tok(7).c_str(), 0, 10) - 1;
	if (mpos < 0 && mtid >= 0) mtid = -1;
	const int32_t isize = (int32_t)strtol(tok(8).c_str(), 0, 10);
	int32_t l_qseq = 0;
	if (!(fe[9] - f[9] == 1 && *f[9] == '*')) {
		l_qseq = (int32_t)(fe[9] - f[9]);
		if (n_cigar && qlen_c != l_qseq) { err = "CIGAR and query sequence are of different length"; return -1; }
		const uint8_t *T = nt16_tab(); const size_t o = out.size();
		out.resize(o + (size_t)((l_qseq + 1) >> 1), 0);
		for (int i = 0; i < l_qseq; ++i) out[o + (size_t)(i >> 1)] |= (uint8_t)(T[(uint8_t)f[9][i]] << ((~i & 1) << 2));
	}
	if (!(fe[10] - f[10] == 1 && *f[10] == '*')) {
		if (fe[10] - f[10] != l_qseq) { err = "SEQ and QUAL are of different length"; return -1; }
		const size_t o = out.size(); out.resize(o + (size_t)l_qseq);
		for (int i = 0; i < l_qseq; ++i) out[o + (size_t)i] = (uint8_t)(f[10][i] - 33);
	} else out.resize(out.size() + (size_t)l_qseq, 0xff);
	for (const char *q = aux; q < e; ) {   /* optional fields */
		const char *t = (const char*)memchr(q, '\t', (size_t)(e - q)); const char *qe = t ? t : e;
		if (qe - q < 6) { err = "incomplete aux field"; return -1; }
		out.push_back((uint8_t)q[0]); out.push_back((uint8_t)q[1]);
		const char type = q[3]; const char *v = q + 5; const std::string val(v, (size_t)(qe - v));
		auto put = [&](const void *x, size_t n) { out.insert(out.end(), (const uint8_t*)x, (const uint8_t*)x + n); };
		if (type == 'A' || type == 'a' || type == 'c' || type == 'C') { out.push_back('A'); out.push_back((uint8_t)(v < qe ? *v : 0)); }
		else if (type == 'i' || type == 'I') {
			if (*v == '-') {
				const long x = strtol(val.c_str(), 0, 10);
				if (x >= INT8_MIN) { out.push_back('c'); out.push_back((uint8_t)(int8_t)x); }
				else if (x >= INT16_MIN) { int16_t y = (int16_t)x; out.push_back('s'); put(&y, 2); }
				else { int32_t y = (int32_t)x; out.push_back('i'); put(&y, 4); }
			} else {
				const unsigned long x = strtoul(val.c_str(), 0, 10);
				if (x <= UINT8_MAX) { out.push_back('C'); out.push_back((uint8_t)x); }
				else if (x <= UINT16_MAX) { uint16_t y = (uint16_t)x; out.push_back('S'); put(&y, 2); }
				else { uint32_t y = (uint32_t)x; out.push_back('I'); put(&y, 4); }
			}
		} else if (type == 'f') { float x = (float)strtod(val.c_str(), 0); out.push_back('f'); put(&x, 4); }
		else if (type == 'd') { double x = strtod(val.c_str(), 0); out.push_back('d'); put(&x, 8); }
		else if (type == 'Z' || type == 'H') { out.push_back((uint8_t)type); put(val.data(), val.size()); out.push_back(0); }
		else if (type == 'B') {
			if (val.size() < 1) { err = "incomplete B-typed aux field"; return -1; }
			const char st = val[0]; int32_t n = 0; for (char c : val) if (c == ',') ++n;
			out.push_back('B'); out.push_back((uint8_t)st); put(&n, 4);
			const char *r = val.c_str() + 1;
			while (*r == ',') {
				char *nx;
				if (st == 'c') { int8_t x = (int8_t)strtol(r + 1, &nx, 0); put(&x, 1); }
				else if (st == 'C') { uint8_t x = (uint8_t)strtoul(r + 1, &nx, 0); put(&x, 1); }
				else if (st == 's') { int16_t x = (int16_t)strtol(r + 1, &nx, 0); put(&x, 2); }
				else if (st == 'S') { uint16_t x = (uint16_t)strtoul(r + 1, &nx, 0); put(&x, 2); }
				else if (st == 'i') { int32_t x = (int32_t)strtol(r + 1, &nx, 0); put(&x, 4); }
				else if (st == 'I') { uint32_t x = (uint32_t)strtoul(r + 1, &nx, 0); put(&x, 4); }
				else if (st == 'f') { float x = (float)strtod(r + 1, &nx); put(&x, 4); }
				else { err = "unrecognized B type"; return -1; }
				r = nx;
			}
		} else { err = "unrecognized aux type"; return -1; }
		q = t ? t + 1 : e;
	}
	uint32_t x[9];
	x[0] = (uint32_t)(out.size() - base - 4);
	x[1] = (uint32_t)tid; x[2] = (uint32_t)pos; x[3] = (uint32_t)bin << 16 | (uint32_t)(mapq & 0xff) << 8 | (uint32_t)l_qname;
	x[4] = (uint32_t)flag << 16 | (uint32_t)n_cigar; x[5] = (uint32_t)l_qseq; x[6] = (uint32_t)mtid; x[7] = (uint32_t)mpos; x[8] = (uint32_t)isize;
	memcpy(out.data() + base, x, 36);
	return 0;
}

/* fields of a BAM record (after the 4-byte block_size) */
struct bam_core_t { int32_t tid, pos; uint32_t bin_mq_nl, flag_nc; int32_t l_qseq, mtid, mpos, isize; };
static inline uint64_t bam_sort_key(const uint8_t *rec)
{	/* bam1_lt's coordinate key */
	bam_core_t c; memcpy(&c, rec, 32);
	return (uint64_t)(int64_t)c.tid << 32 | (uint64_t)(uint32_t)((c.pos + 1) << 1) | ((c.flag_nc >> 16) & 0x10 ? 1u : 0u);
}
static inline int32_t bam_endpos(const uint8_t *rec)
{
	bam_core_t c; memcpy(&c, rec, 32);
	const uint32_t flag = c.flag_nc >> 16, n_cigar = c.flag_nc & 0xffff, l_qname = c.bin_mq_nl & 0xff;
	if ((flag & 4) || n_cigar == 0) return c.pos + 1;
	int32_t rl = 0;
	for (uint32_t i = 0; i < n_cigar; ++i) { uint32_t v; memcpy(&v, rec + 32 + l_qname + 4 * i, 4); const uint32_t op = v & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += (int32_t)(v >> 4); }
	return c.pos + rl;
}

/* ---- BAI (hts_idx_*) ---- */
struct bai_t {
	struct chunk_t { uint64_t u, v; };
	struct ref_t { std::map<uint32_t, std::vector<chunk_t> > bins; std::vector<uint64_t> lin; bool seen; ref_t() : seen(false) {} };
	std::vector<ref_t> refs; uint64_t n_no_coor;
	uint32_t last_bin, save_bin; int last_coor, last_tid, save_tid; uint64_t last_off, save_off, off_beg, off_end, n_mapped, n_unmapped;
	bai_t(int n_ref, uint64_t off0) : refs((size_t)n_ref), n_no_coor(0), last_bin(0xffffffffu), save_bin(0xffffffffu), last_coor((int)0xffffffffu),
		last_tid((int)0xffffffffu), save_tid((int)0xffffffffu), last_off(off0), save_off(off0), off_beg(off0), off_end(off0), n_mapped(0), n_unmapped(0) {}
	void to_b(int tid, uint32_t bin, uint64_t u, uint64_t v) { chunk_t c; c.u = u; c.v = v; refs[(size_t)tid].bins[bin].push_back(c); refs[(size_t)tid].seen = true; }
	int push(int tid, int beg, int end, uint64_t offset, bool mapped)
	{
		if (tid < 0) { beg = -1; end = 0; }
		if (tid >= (int)refs.size()) refs.resize((size_t)tid + 1);
		if (last_tid != tid || (last_tid >= 0 && tid < 0)) {
			if (tid >= 0 && n_no_coor) return -1;
			if (tid >= 0 && refs[(size_t)tid].seen) return -1;
			last_tid = tid; last_bin = 0xffffffffu;
		} else if (tid >= 0 && last_coor > beg) return -1;
		if (tid >= 0) {
			refs[(size_t)tid].seen = true;
			if (mapped) {   /* insert_to_l with last_off = start of this record */
				std::vector<uint64_t> &l = refs[(size_t)tid].lin; const int b = beg >> 14, e = (end - 1) >> 14;
				if ((int)l.size() < e + 1) l.resize((size_t)e + 1, (uint64_t)-1);
				for (int i = b; i <= e; ++i) if (l[(size_t)i] == (uint64_t)-1) l[(size_t)i] = last_off;
			}
		} else ++n_no_coor;
		const uint32_t bin = (uint32_t)reg2bin(beg, end);
		if (last_bin != bin) {
			if (save_bin != 0xffffffffu) to_b(save_tid, save_bin, save_off, last_off);
			if (last_bin == 0xffffffffu && save_bin != 0xffffffffu) {
				off_end = last_off;
				to_b(save_tid, 37450, off_beg, off_end); to_b(save_tid, 37450, n_mapped, n_unmapped);
				n_mapped = n_unmapped = 0; off_beg = off_end;
			}
			save_off = last_off; save_bin = last_bin = bin; save_tid = tid;
		}
		if (mapped) ++n_mapped; else ++n_unmapped;
		last_off = offset; last_coor = beg;
		return 0;
	}
	void finish(uint64_t final_offset)
	{
		if (save_tid >= 0) { to_b(save_tid, save_bin, save_off, final_offset); to_b(save_tid, 37450, off_beg, final_offset); to_b(save_tid, 37450, n_mapped, n_unmapped); }
		for (ref_t &R : refs) {
			/* update_loff: leading / missing linear-index entries */
			uint64_t off0 = 0; auto mk = R.bins.find(37450);
			if (!R.bins.empty()) { if (mk != R.bins.end()) off0 = mk->second[0].u; size_t l = 0; for (; l < R.lin.size() && R.lin[l] == (uint64_t)-1; ++l) R.lin[l] = off0; }
			for (size_t l = 1; l < R.lin.size(); ++l) if (R.lin[l] == (uint64_t)-1) R.lin[l] = R.lin[l - 1];
			if (R.bins.empty()) continue;
			/* compress_binning */
			auto by_u = [](const chunk_t &a, const chunk_t &b) { return a.u < b.u; };
			for (int l = 5; l > 0; --l) {
				const uint32_t start = (uint32_t)(((1 << ((l << 1) + l)) - 1) / 7);
				for (auto it = R.bins.lower_bound(start); it != R.bins.end(); ) {
					if (it->first >= 37449) { ++it; continue; }
					std::vector<chunk_t> &p = it->second;
					if (l < 5 && p.size() > 1) std::sort(p.begin(), p.end(), by_u);
					if ((p.back().v >> 16) - (p.front().u >> 16) < 0x10000) {
						auto par = R.bins.find((it->first - 1) >> 3);
						if (par == R.bins.end()) { ++it; continue; }
						par->second.insert(par->second.end(), p.begin(), p.end());
						it = R.bins.erase(it);
					} else ++it;
				}
			}
			auto b0 = R.bins.find(0);
			if (b0 != R.bins.end()) std::sort(b0->second.begin(), b0->second.end(), by_u);
			for (auto &kv : R.bins) {
				if (kv.first >= 37449) continue;
				std::vector<chunk_t> &p = kv.second; size_t m = 0;
				for (size_t l = 1; l < p.size(); ++l) { if (p[m].v >> 16 >= p[l].u >> 16) { if (p[m].v < p[l].v) p[m].v = p[l].v; } else p[++m] = p[l]; }
				p.resize(m + 1);
			}
		}
	}
	void save(const char *fn) const
	{
		FILE *fp = fopen(fn, "wb"); if (!fp) { perror(fn); exit(1); }
		auto w32 = [&](int32_t v) { fwrite(&v, 4, 1, fp); }; auto w64 = [&](uint64_t v) { fwrite(&v, 8, 1, fp); };
		fwrite("BAI\1", 1, 4, fp); w32((int32_t)refs.size());
		for (const ref_t &R : refs) {
			w32((int32_t)R.bins.size());
			for (const auto &kv : R.bins) { w32((int32_t)kv.first); w32((int32_t)kv.second.size()); for (const chunk_t &c : kv.second) { w64(c.u); w64(c.v); } }
			w32((int32_t)R.lin.size()); for (uint64_t o : R.lin) w64(o);
		}
		w64(n_no_coor);
		fclose(fp);
	}
};
#endif
