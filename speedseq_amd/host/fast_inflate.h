/*
 * fast_inflate.h -- gzip (RFC 1952 / 1951) decoder for `bwa mem`'s FASTQ ingest.
 *
 * `.fastq.gz` is the reference's normal input (/root/reference/bin/speedseq:199-200, example/run_speedseq.sh:4-10) and one zlib
 * inflate stream (~190 MB/s of text on the build host, ~350 on the GPU box) is what an interleaved gz file can deliver: a quarter of
 * what one MI355X aligns.  This decoder does the same job with a 64-bit bit buffer that is topped up with one unaligned load, table
 * look-ups that resolve literal / length codes of up to 11 bits and distance codes of up to 8 bits in one step (longer codes through a
 * second-level table), and match copies eight bytes at a time.  The CRC-32 of RFC 1952 is checked by the caller's thread of choice
 * (fq_stream_t gives it a thread of its own: zlib's crc32 is not faster than this decoder).  Output is byte for byte what zlib's
 * inflate produces (tests/test_fast_inflate.py: every compression level, stored / fixed / dynamic blocks, several members,
 * truncated and damaged streams); SSG_GZ_FAST=0 puts zlib back.
 *
 * Not a general library: sequential decoding of a whole file, output in caller-sized chunks, errors are final.
 */
#ifndef SSG_FAST_INFLATE_H
#define SSG_FAST_INFLATE_H
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <vector>

struct fast_gz_t {
	enum { PB_LIT = 11, PB_DIST = 8, WIN = 32768, SLACK = 512 };
	/* table entry: bits 0-7 bits to drop, 8-15 sub-table bits (link; lengths and distances: see mkbase), 16-29 value (literal, length
	 * base, distance symbol, or sub-table offset), 30-31 kind */
	enum { K_LIT = 0u, K_BASE = 1u, K_EOB = 2u, K_LINK = 3u };
	static inline uint32_t mk(uint32_t kind, uint32_t val, uint32_t extra, uint32_t nbits) { return kind << 30 | val << 16 | extra << 8 | nbits; }
	/* length / distance entries drop code and extra bits with ONE shift (the serial chain through the bit buffer is what bounds the
	 * decoder): bits 0-7 code + extra bits, 8-11 code bits, 12-15 extra bits */
	static inline uint32_t mkbase(uint32_t val, uint32_t extra, uint32_t nbits) { return K_BASE << 30 | val << 16 | extra << 12 | nbits << 8 | (nbits + extra); }
	static const uint32_t BAD = 0xffffffffu;

	int fd; bool eof_in;                                   /* input file, and whether read() has returned 0 */
	std::vector<uint8_t> ibuf; const uint8_t *ib; size_t ip, iend, ireal;   /* ib[ip..iend) unread (ib = ibuf, or a caller's buffer: fast_inflate_mt.h); iend may include zero padding after the file's end (ireal) */
	uint64_t bitbuf; int bitcnt;
	std::vector<uint8_t> obuf; size_t op, obase;           /* obuf[obase..op) = decoded and not yet handed out; WIN bytes of history before obase */
	enum { S_HEADER, S_BLOCK, S_STORED, S_HUFF, S_TRAILER, S_DONE, S_ERROR } st;
	bool last_block; uint32_t stored_left;
	std::vector<uint32_t> tl, td;                          /* litlen / dist tables (primary + sub-tables) */
	uint32_t crc_expect, isize_expect; uint64_t member_out; bool member_done;   /* set when a member's trailer has been read */
	const char *err;
	bool strict; int last_left, last_maxl;                            /* strict: a dynamic block header must describe complete codes (the block-boundary search of fast_inflate_mt.h) */

	explicit fast_gz_t(int fd_) : fd(fd_), eof_in(false), ib(0), ip(0), iend(0), ireal(0), bitbuf(0), bitcnt(0), op(WIN), obase(WIN), st(S_HEADER), last_block(false),
		stored_left(0), crc_expect(0), isize_expect(0), member_out(0), member_done(false), err(0), strict(false), last_left(0), last_maxl(0)
	{ if (fd >= 0) ibuf.resize((size_t)4 << 20); ib = ibuf.data(); tl.resize(((size_t)1 << PB_LIT) + 2048); td.resize(((size_t)1 << PB_DIST) + 2048); }

	/* ---- input ---- */
	bool fill_input()
	{	/* keep at least 4 KB ahead (a dynamic block header is < 1 KB); after the file's end the buffer is padded with zeros so that the
		 * decoder's unconditional 8-byte loads stay inside it -- consumption past `ireal` is reported as truncation */
		if (iend - ip >= 8192) return true;
		if (fd < 0 || ib != ibuf.data()) return true;   /* a caller's buffer (fast_inflate_mt.h parses member headers that way): nothing to fill or move; what is missing shows as ip >= ireal */
		{ const int held = bitcnt >> 3; ip -= (size_t)held; bitcnt &= 7; bitbuf &= ((uint64_t)1 << bitcnt) - 1; }   /* whole bytes waiting in the bit buffer go back: the move below must not lose them */
		if (ip) { memmove(ibuf.data(), ibuf.data() + ip, iend - ip); iend -= ip; if (ireal >= ip) ireal -= ip; else ireal = 0; ip = 0; }
		while (!eof_in && iend < ibuf.size() - 4096) {
			const ssize_t r = read(fd, ibuf.data() + iend, ibuf.size() - 4096 - iend);
			if (r < 0) { if (errno == EINTR) continue; err = "read error"; return false; }
			if (r == 0) { eof_in = true; break; }
			iend += (size_t)r; ireal = iend;
		}
		if (eof_in && iend == ireal) { const size_t pad = std::min<size_t>(4096, ibuf.size() - iend); memset(ibuf.data() + iend, 0, pad); iend += pad; }
		return true;
	}
	inline void refill() { uint64_t w; memcpy(&w, ib + ip, 8); bitbuf |= w << bitcnt; ip += (size_t)((63 - bitcnt) >> 3); bitcnt |= 56; }
	inline uint32_t bits(int n) { const uint32_t v = (uint32_t)(bitbuf & (((uint64_t)1 << n) - 1)); bitbuf >>= n; bitcnt -= n; return v; }
	size_t consumed() const { return ip - (size_t)(bitcnt >> 3); }      /* input offset of the first byte not (wholly) used */
	void byte_align() { const int drop = bitcnt & 7; bitbuf >>= drop; bitcnt -= drop; ip -= (size_t)(bitcnt >> 3); bitbuf = 0; bitcnt = 0; }

	/* ---- tables ---- */
	static uint32_t rev(uint32_t c, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) { r = r << 1 | (c & 1); c >>= 1; } return r; }
	/* canonical Huffman code of `lens` (n symbols, lengths 0..15) -> look-up table; entry of symbol s is ent(s, bits to drop) */
	template <class F> bool build(const uint8_t *lens, int n, int pb, std::vector<uint32_t> &t, F ent)
	{
		int cnt[16] = { 0 }; for (int s = 0; s < n; ++s) ++cnt[lens[s]];
		cnt[0] = 0;
		int left = 1, maxl = 0;
		for (int l = 1; l <= 15; ++l) { left = (left << 1) - cnt[l]; if (left < 0) return false; if (cnt[l]) maxl = l; }   /* over-subscribed */
		last_left = left; last_maxl = maxl;                                 /* left 0: the code is complete */
		if (maxl == 0) { for (size_t i = 0; i < ((size_t)1 << pb); ++i) t[i] = BAD; return true; }                       /* no codes: any use is an error */
		uint32_t next[16]; { uint32_t c = 0; for (int l = 1; l <= 15; ++l) { c = (c + (uint32_t)cnt[l - 1]) << 1; next[l] = c; } }
		for (size_t i = 0; i < ((size_t)1 << pb); ++i) t[i] = BAD;
		/* sub-table sizes: the longest code under each primary prefix */
		uint8_t sub[1 << PB_LIT]; memset(sub, 0, (size_t)1 << pb);
		uint32_t code_of[320];
		{ uint32_t nx[16]; memcpy(nx, next, sizeof(nx)); for (int s = 0; s < n; ++s) { const int l = lens[s]; if (!l) continue; const uint32_t r = rev(nx[l]++, l); code_of[s] = r; if (l > pb) { uint8_t &b = sub[r & (((uint32_t)1 << pb) - 1)]; if (l - pb > b) b = (uint8_t)(l - pb); } } }
		size_t used = (size_t)1 << pb;
		for (size_t p = 0; p < ((size_t)1 << pb); ++p) if (sub[p]) {
			if (used + ((size_t)1 << sub[p]) > t.size()) t.resize(used + ((size_t)1 << sub[p]) + 1024);
			t[p] = mk(K_LINK, (uint32_t)used, sub[p], (uint32_t)pb);
			for (size_t k = 0; k < ((size_t)1 << sub[p]); ++k) t[used + k] = BAD;
			used += (size_t)1 << sub[p];
		}
		if (used > (1u << 14)) return false;
		for (int s = 0; s < n; ++s) {
			const int l = lens[s]; if (!l) continue;
			const uint32_t r = code_of[s];
			if (l <= pb) { const uint32_t e = ent(s, (uint32_t)l); for (uint32_t k = r; k < ((uint32_t)1 << pb); k += (uint32_t)1 << l) t[k] = e; }
			else {
				const uint32_t p = r & (((uint32_t)1 << pb) - 1), sb = sub[p], off = (t[p] >> 16) & 0x3fff, e = ent(s, (uint32_t)(l - pb));
				for (uint32_t k = r >> pb; k < ((uint32_t)1 << sb); k += (uint32_t)1 << (l - pb)) t[off + k] = e;
			}
		}
		return true;
	}
	static uint32_t ent_lit(int s, uint32_t nb)
	{
		static const uint16_t base[29] = { 3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258 };
		static const uint8_t extra[29] = { 0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0 };
		if (s < 256) return mk(K_LIT, (uint32_t)s, 0, nb);
		if (s == 256) return mk(K_EOB, 0, 0, nb);
		if (s > 285) return BAD;
		return mkbase(base[s - 257], extra[s - 257], nb);
	}
	static uint32_t ent_dist(int s, uint32_t nb)
	{
		static const uint16_t base[30] = { 1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577 };
		static const uint8_t extra[30] = { 0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13 };
		if (s > 29) return BAD;
		/* the base does not fit the 14-bit value field: the symbol goes there, the base is looked up when the entry is used */
		(void)base;
		return mkbase((uint32_t)s, extra[s], nb);
	}
	static inline uint32_t dist_base(uint32_t s)
	{
		static const uint16_t base[30] = { 1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577 };
		return base[s];
	}
	bool fixed_tables()
	{
		uint8_t l[288]; for (int i = 0; i < 144; ++i) l[i] = 8; for (int i = 144; i < 256; ++i) l[i] = 9; for (int i = 256; i < 280; ++i) l[i] = 7; for (int i = 280; i < 288; ++i) l[i] = 8;
		uint8_t d[30]; memset(d, 5, 30);
		return build(l, 288, PB_LIT, tl, ent_lit) && build(d, 30, PB_DIST, td, ent_dist);
	}
	bool dynamic_tables()
	{	/* RFC 1951 3.2.7; the caller made sure a whole header's worth of input is in the buffer */
		refill();
		const int hlit = (int)bits(5) + 257, hdist = (int)bits(5) + 1, hclen = (int)bits(4) + 4;
		if (hlit > 286 || hdist > 30) return false;
		static const uint8_t order[19] = { 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 };
		uint8_t cl[19]; memset(cl, 0, 19);
		for (int i = 0; i < hclen; ++i) { if (bitcnt < 3) refill(); cl[order[i]] = (uint8_t)bits(3); }
		std::vector<uint32_t> tc(((size_t)1 << 7) + 64);
		if (!build(cl, 19, 7, tc, [](int s, uint32_t nb) { return mk(K_LIT, (uint32_t)s, 0, nb); }) || last_left != 0) return false;   /* zlib (inftrees.c): the code-length code must be complete */
		uint8_t lens[320]; int i = 0;
		while (i < hlit + hdist) {
			refill();
			const uint32_t e = tc[bitbuf & 127];
			if (e == BAD || (e >> 30) != K_LIT) return false;
			(void)bits((int)(e & 0xff));
			const int s = (int)((e >> 16) & 0x3fff);
			if (s < 16) { lens[i++] = (uint8_t)s; continue; }
			int rep; uint8_t v = 0;
			if (s == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + (int)bits(2); }
			else if (s == 17) rep = 3 + (int)bits(3);
			else rep = 11 + (int)bits(7);
			if (i + rep > hlit + hdist) return false;
			while (rep--) lens[i++] = v;
		}
		if (lens[256] == 0) return false;                                  /* no end-of-block code */
		/* zlib's rule (inftrees.c: `left > 0 && (type == CODES || max != 1)' is an error): a literal / length or distance code may be incomplete
		 * only when it consists of a single one-bit code; the boundary search of fast_inflate_mt.h (strict) takes no incomplete literal code at all */
		if (!build(lens, hlit, PB_LIT, tl, ent_lit) || (last_left != 0 && (strict || last_maxl != 1))) return false;
		if (!build(lens + hlit, hdist, PB_DIST, td, ent_dist) || (last_left != 0 && last_maxl > 1)) return false;
		return true;
	}

	/* the symbols of a Huffman block into ob[o..): stops at `limit`, at the end of the buffered input (both 0), at the end of the block (1),
	 * on damage (-1, err).  E = uint8_t, or uint16_t when the 32 KB before the start are not known yet and stand in the buffer as markers
	 * (fast_inflate_mt.h); `avail` = elements before o that a distance may reach.  Works on local copies of the bit buffer: the stores to
	 * the output may alias anything the compiler cannot see through. */
	template <class E> int huff_run(E *const ob, size_t &o_io, const size_t limit, const uint64_t avail)
	{
		const uint32_t *const L = tl.data(), *const D = td.data(); const uint8_t *const ibl = ib;
		size_t o = o_io, ipl = ip; const size_t o0 = o_io;
		uint64_t bb = bitbuf; int bc = bitcnt;
		const size_t in_stop = iend >= 16 ? iend - 16 : 0;              /* 8-byte loads stay inside the (padded) buffer */
		const size_t EW = 8 / sizeof(E);                                /* elements per 8-byte word */
		int ret = 0; const char *bad = 0;
#define FI_REFILL() do { uint64_t w_; memcpy(&w_, ibl + ipl, 8); bb |= w_ << bc; ipl += (size_t)((63 - bc) >> 3); bc |= 56; } while (0)
		while (o < limit && ipl < in_stop) {
			FI_REFILL();
			uint32_t e = L[bb & ((1u << PB_LIT) - 1)];
			if ((e >> 30) == K_LINK) { if (e == BAD) { bad = "damaged literal / length code"; break; } bb >>= PB_LIT; bc -= PB_LIT; e = L[((e >> 16) & 0x3fff) + (bb & ((1u << ((e >> 8) & 0xff)) - 1))]; if (e == BAD) { bad = "damaged literal / length code"; break; } }
			uint64_t sv = bb;
			bb >>= (e & 0xff); bc -= (int)(e & 0xff);
			const uint32_t kind = e >> 30;
			if (kind == K_LIT) {
				ob[o++] = (E)(uint8_t)(e >> 16);
				/* up to three more literals from the same top-up (each <= 11 bits, the first symbol took <= 15) */
				e = L[bb & ((1u << PB_LIT) - 1)];
				if ((e >> 30) != K_LIT) continue;
				bb >>= (e & 0xff); bc -= (int)(e & 0xff); ob[o++] = (E)(uint8_t)(e >> 16);
				e = L[bb & ((1u << PB_LIT) - 1)];
				if ((e >> 30) != K_LIT) continue;
				bb >>= (e & 0xff); bc -= (int)(e & 0xff); ob[o++] = (E)(uint8_t)(e >> 16);
				e = L[bb & ((1u << PB_LIT) - 1)];
				if ((e >> 30) != K_LIT) continue;
				bb >>= (e & 0xff); bc -= (int)(e & 0xff); ob[o++] = (E)(uint8_t)(e >> 16);
				continue;
			}
			if (kind == K_EOB) { ret = 1; break; }
			/* length (code + extra bits already dropped), then distance: at most 15 + 13 further bits; the top-up left at least 56 - 20 */
			uint32_t len = ((e >> 16) & 0x3fff) + (uint32_t)((sv >> ((e >> 8) & 0xf)) & ((1u << ((e >> 12) & 0xf)) - 1));
			uint32_t d = D[bb & ((1u << PB_DIST) - 1)];
			if ((d >> 30) == K_LINK) { if (d == BAD) { bad = "damaged distance code"; break; } bb >>= PB_DIST; bc -= PB_DIST; d = D[((d >> 16) & 0x3fff) + (bb & ((1u << ((d >> 8) & 0xff)) - 1))]; }
			if ((d >> 30) != K_BASE) { bad = "damaged distance code"; break; }
			sv = bb;
			bb >>= (d & 0xff); bc -= (int)(d & 0xff);
			const size_t dist = dist_base((d >> 16) & 0x3fff) + (size_t)((sv >> ((d >> 8) & 0xf)) & (((uint64_t)1 << ((d >> 12) & 0xf)) - 1));
			if (dist > avail + (o - o0) || dist > WIN) { bad = "distance too far back"; break; }
			const E *sp = ob + o - dist; E *t = ob + o;
			o += len;
			if (__builtin_expect(dist >= EW, 1)) {                      /* a word at once never reads what it is about to write */
				uint64_t w; memcpy(&w, sp, 8); memcpy(t, &w, 8);
				if (__builtin_expect(len > EW, 0)) { sp += EW; t += EW; do { memcpy(&w, sp, 8); memcpy(t, &w, 8); sp += EW; t += EW; } while (t < ob + o); }
			}
			else if (dist == 1) { const E v = sp[0]; while (len--) *t++ = v; }
			else { while (len--) *t++ = *sp++; }
		}
#undef FI_REFILL
		bitbuf = bb; bitcnt = bc; ip = ipl; o_io = o;
		if (bad) { err = bad; return -1; }
		return ret;
	}

	/* ---- gzip member header / trailer (byte-aligned, through the same buffer) ---- */
	int byte() { if (ip >= ireal) return -1; return ib[ip++]; }
	bool header()
	{
		if (!fill_input()) return false;
		if (ip >= ireal) { st = S_DONE; return true; }                     /* clean end between members */
		if (ireal - ip < 18 && !eof_in) { err = "short header"; return false; }
		const int a = byte(), b = byte(), cm = byte(), flg = byte();
		if (a != 0x1f || b != 0x8b) {
			if (member_done) { st = S_DONE; return true; }                  /* trailing garbage after a complete member is ignored, as gzread does */
			err = "not a gzip stream"; return false;
		}
		if (cm != 8 || (flg & 0xe0)) { err = "unsupported gzip header"; return false; }
		for (int i = 0; i < 6; ++i) if (byte() < 0) { err = "truncated gzip header"; return false; }
		if (flg & 4) { const int lo = byte(), hi = byte(); if (hi < 0) { err = "truncated gzip header"; return false; } size_t n = (size_t)lo | (size_t)hi << 8;
			while (n) { if (!fill_input()) return false; if (ip >= ireal) { err = "truncated gzip header"; return false; } const size_t k = std::min(n, ireal - ip); ip += k; n -= k; } }
		for (int f = 8; f <= 16; f <<= 1) if (flg & f) for (;;) { if (ip >= ireal) { if (!fill_input()) return false; if (ip >= ireal) { err = "truncated gzip header"; return false; } } if (ib[ip++] == 0) break; }
		if (flg & 2) { if (byte() < 0 || byte() < 0) { err = "truncated gzip header"; return false; } }
		bitbuf = 0; bitcnt = 0; member_out = 0; member_done = false;
		st = S_BLOCK;
		return true;
	}

	/* decodes until at least `want` bytes are ready, a member ends or the stream does: *p .. *p + return value (valid until the next call);
	 * *member_end = the chunk ends a gzip member (crc_expect is the CRC-32 of everything returned since the previous one); 0 without
	 * *member_end = end of stream; -1 = error (err) */
	long read_chunk(const uint8_t **p, size_t want, bool *member_end)
	{
		if (st == S_ERROR) return -1;
		/* slide: keep the last WIN bytes as history in front of the new output */
		if (obuf.size() < WIN + want + 65536 + SLACK) obuf.resize(WIN + want + 65536 + SLACK);
		if (op != WIN) { memmove(obuf.data(), obuf.data() + op - WIN, WIN); op = obase = WIN; }   /* op >= WIN always: the first WIN bytes are history (valid as far as member_out says) */
		*member_end = false;
		const size_t limit = WIN + want;
		while (op < limit && st != S_DONE) {
			if (!fill_input()) { st = S_ERROR; return -1; }
			switch (st) {
			case S_HEADER: if (!header()) { st = S_ERROR; return -1; } break;
			case S_BLOCK: {
				refill();
				last_block = bits(1) != 0;
				const uint32_t type = bits(2);
				if (type == 0) {
					byte_align();
					if (ireal - ip < 4) { err = "truncated stored block"; st = S_ERROR; return -1; }
					const uint32_t len = ib[ip] | (uint32_t)ib[ip + 1] << 8, nlen = ib[ip + 2] | (uint32_t)ib[ip + 3] << 8;
					if ((len ^ 0xffffu) != nlen) { err = "damaged stored block"; st = S_ERROR; return -1; }
					ip += 4; stored_left = len; st = S_STORED;
				} else if (type == 1) { if (!fixed_tables()) { err = "internal: fixed tables"; st = S_ERROR; return -1; } st = S_HUFF; }
				else if (type == 2) { if (!dynamic_tables()) { err = "damaged dynamic block header"; st = S_ERROR; return -1; } st = S_HUFF; }
				else { err = "reserved block type"; st = S_ERROR; return -1; }
			} break;
			case S_STORED: {
				size_t k = std::min<size_t>(stored_left, ireal > ip ? ireal - ip : 0);
				k = std::min(k, obuf.size() - SLACK - op);
				if (!k && stored_left) { if (ip >= ireal && eof_in) { err = "truncated stored block"; st = S_ERROR; return -1; } if (op >= limit) break; }
				memcpy(obuf.data() + op, ib + ip, k); op += k; ip += k; stored_left -= (uint32_t)k; member_out += k;
				if (!stored_left) st = last_block ? S_TRAILER : S_BLOCK;
			} break;
			case S_HUFF: {
				size_t o = op; const size_t o0 = op;
				const int r = huff_run<uint8_t>(obuf.data(), o, limit, member_out);
				member_out += o - o0; op = o;
				if (r < 0) { st = S_ERROR; return -1; }
				if (r == 1) st = last_block ? S_TRAILER : S_BLOCK;
				else if (consumed() > ireal) { err = "truncated deflate stream"; st = S_ERROR; return -1; }
			} break;
			case S_TRAILER: {
				byte_align();
				if (ip > ireal) { err = "truncated deflate stream"; st = S_ERROR; return -1; }
				if (ireal - ip < 8) { err = "truncated gzip trailer"; st = S_ERROR; return -1; }
				memcpy(&crc_expect, ib + ip, 4); memcpy(&isize_expect, ib + ip + 4, 4); ip += 8;
				if ((uint32_t)member_out != isize_expect) { err = "length in the gzip trailer does not match"; st = S_ERROR; return -1; }
				member_done = true; st = S_HEADER;
				/* the caller checks crc_expect against the CRC-32 of the bytes of this member: a chunk never spans two members */
				*member_end = true;
				*p = obuf.data() + obase; const long n = (long)(op - obase); obase = op; return n;
			}
			default: break;
			}
		}
		*p = obuf.data() + obase; const long n = (long)(op - obase); obase = op;
		return n;
	}
};
#endif
