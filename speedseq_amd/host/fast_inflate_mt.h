/*
 * fast_inflate_mt.h -- ONE gzip stream decoded by several threads (bwa mem's FASTQ ingest).
 *
 * A deflate stream can only be entered at a block boundary, and what a block's matches copy from may lie in the 32 KB before it.
 * Both are dealt with the way pugz / rapidgzip do it:
 *   - the compressed bytes are taken in waves of T chunks; a thread that does not hold the stream's true position looks for a
 *     non-final dynamic block header at or after its chunk's first byte (bit by bit: header fields in range, the code-length code
 *     complete, both symbol codes complete, an end-of-block code present);
 *   - from there it decodes into 16-bit elements: a byte, or -- for a match that reaches behind the chunk's start -- a marker naming
 *     the position in the unknown 32 KB window; it stops at a block boundary where the next chunk started (so the pieces abut
 *     exactly), or carries on over a chunk whose "boundary" turns out not to be one;
 *   - the chunks are then walked in stream order: the window before each is known by then, markers become bytes (all threads), the
 *     bytes go to the caller, and the last 32 KB become the next window.
 * A false block boundary costs only the work of the thread that believed it: nobody lands on it, its output is dropped.  Whatever
 * does not fit the scheme (a block longer than the slack kept behind a wave, a stream of stored blocks, damage) hands the rest of the
 * member to the one-thread decoder at the exact bit where the last good chunk ended, with its window.  The CRC-32 of every member is
 * computed by the same threads that turn markers into bytes (slice by slice, joined with crc32_combine) and checked here against the
 * trailer, as is the length.
 */
#ifndef SSG_FAST_INFLATE_MT_H
#define SSG_FAST_INFLATE_MT_H
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <sys/stat.h>
#include <atomic>
#include <vector>
#include <thread>
#include <functional>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <algorithm>
#include <zlib.h>   /* crc32, crc32_combine */
#include "fast_inflate.h"

struct fast_gz_mt_t {
	enum { WIN = fast_gz_t::WIN };
	int fd, T; size_t C;                                   /* threads; compressed bytes per chunk */
	const char *err; uint64_t file_size;
	/* sink(bytes, n, member_end, crc_expect): bytes of the stream in order; member_end closes a gzip member (its CRC-32 and length have
	 * been checked here by then) */
	typedef std::function<bool(const uint8_t*, size_t, bool, uint32_t)> sink_t;
	fast_gz_mt_t(int fd_, int threads, size_t chunk = (size_t)2 << 20) : fd(fd_), T(threads < 2 ? 2 : threads), C(chunk), err(0), file_size(0) {}

	static bool pread_all(int fd, void *buf, size_t n, uint64_t off)
	{
		uint8_t *b = (uint8_t*)buf;
		while (n) { const ssize_t r = pread(fd, b, n, (off_t)off); if (r < 0) { if (errno == EINTR) continue; return false; } if (r == 0) return false; b += r; n -= (size_t)r; off += (uint64_t)r; }
		return true;
	}
	/* a decoder over a caller's buffer, positioned at bit `bit` of it */
	static void place(fast_gz_t &g, const uint8_t *buf, size_t real, size_t padded, uint64_t bit)
	{
		g.ib = buf; g.ireal = real; g.iend = padded; g.eof_in = true; g.ip = (size_t)(bit >> 3); g.bitbuf = 0; g.bitcnt = 0;
		g.refill(); (void)g.bits((int)(bit & 7));
	}
	static uint64_t bitpos(const fast_gz_t &g) { return (uint64_t)g.ip * 8 - (uint64_t)g.bitcnt; }

	/* first bit position in [from, to) (bits of buf) where a non-final dynamic block with complete codes starts; -1: none */
	static int64_t find_block(const uint8_t *buf, size_t real, size_t padded, uint64_t from, uint64_t to)
	{
		fast_gz_t g(-1);
		for (uint64_t p = from; p < to; ++p) {
			if ((p >> 3) + 1024 > real) break;                              /* a header may take ~600 bytes; the tail of a wave is the previous chunk's */
			uint64_t w; memcpy(&w, buf + (p >> 3), 8); w >>= (p & 7);
			if ((w & 7) != 4) continue;                                    /* BFINAL = 0, BTYPE = 10 */
			const unsigned hlit = (unsigned)(w >> 3) & 31, hdist = (unsigned)(w >> 8) & 31, hclen = ((unsigned)(w >> 13) & 15) + 4;
			if (hlit > 29 || hdist > 29) continue;
			/* the code-length code must be complete: sum over its lengths of 2^(7 - len) = 128 */
			uint64_t q = p + 17; unsigned kraft = 0; bool ok = true;
			for (unsigned i = 0; i < hclen; ++i, q += 3) { uint16_t v; memcpy(&v, buf + (q >> 3), 2); const unsigned l = (v >> (q & 7)) & 7; if (l) kraft += 128u >> l; if (kraft > 128) { ok = false; break; } }
			if (!ok || kraft != 128) continue;
			place(g, buf, real, padded, p + 3);
			g.strict = true;
			if (g.dynamic_tables()) return (int64_t)p;
		}
		return -1;
	}

	struct chunk_t {
		uint64_t start, end; bool valid, final, failed; std::vector<uint16_t> out; size_t n;   /* out[WIN .. WIN + n): elements; out[0..WIN): markers */
		chunk_t() : start(0), end(0), valid(false), final(false), failed(false), n(0) {}
	};
	/* decodes blocks from c.start (a bit of buf) until a block ends on one of `stops` (sorted bit positions of later chunks), at or after
	 * `soft_end`, or with the final block; c.failed when the buffer runs out first or the data is damaged */
	static void decode_chunk(chunk_t &c, const uint8_t *buf, size_t real, size_t padded, const std::vector<uint64_t> &stops, uint64_t soft_end, bool to_eof)
	{
		fast_gz_t g(-1);
		place(g, buf, real, padded, c.start);
		if (c.out.size() < WIN + ((size_t)8 << 20)) c.out.resize(WIN + ((size_t)8 << 20));   /* kept from wave to wave: no fresh pages, no zero fill */
		for (size_t i = 0; i < (size_t)WIN; ++i) c.out[i] = (uint16_t)(0x8000u | i);
		size_t o = WIN;
		for (;;) {
			if (!to_eof && g.ip + 1024 > real) { c.failed = true; break; }   /* a block header may need ~600 bytes; at the file's end the zero padding stands in */
			g.refill();
			const bool last = g.bits(1) != 0; const uint32_t type = g.bits(2);
			if (type == 0) {
				g.byte_align();
				if (g.ip + 4 > real) { c.failed = true; break; }
				const uint32_t len = buf[g.ip] | (uint32_t)buf[g.ip + 1] << 8, nlen = buf[g.ip + 2] | (uint32_t)buf[g.ip + 3] << 8;
				if ((len ^ 0xffffu) != nlen || g.ip + 4 + len > real) { c.failed = true; break; }
				g.ip += 4;
				if (o + len + 512 > c.out.size()) c.out.resize(c.out.size() * 2 + len);
				for (uint32_t k = 0; k < len; ++k) c.out[o + k] = buf[g.ip + k];
				o += len; g.ip += len;
			} else if (type == 3) { c.failed = true; break; }
			else {
				if (!(type == 1 ? g.fixed_tables() : g.dynamic_tables())) { c.failed = true; break; }
				int r = 0;
				for (;;) {
					if (o + 65536 > c.out.size()) c.out.resize(c.out.size() * 2);
					r = g.huff_run<uint16_t>(c.out.data(), o, c.out.size() - 1024, (uint64_t)1 << 40);
					if (r != 0) break;
					if (o + 65536 > c.out.size()) continue;                   /* stopped for room */
					r = -1; break;                                            /* stopped because the buffered input ran out inside the block */
				}
				if (r < 0 || g.consumed() > real) { c.failed = true; break; }
			}
			const uint64_t here = bitpos(g);
			if (last) { c.final = true; c.end = here; break; }
			if (std::binary_search(stops.begin(), stops.end(), here) || here >= soft_end) { c.end = here; break; }
		}
		c.n = o - WIN;
		if (c.failed) c.n = 0;
	}

	template <class F> void parallel(int n, F f) { std::vector<std::thread> th; for (int t = 1; t < n; ++t) th.emplace_back(f, t); f(0); for (auto &x : th) x.join(); }

	/* one wave as the producer hands it to the stitcher: the chunks that abut, in stream order, and what ends with them */
	struct wave_t {
		uint64_t A; std::vector<chunk_t> ch; std::vector<int> chain; int fail_idx;     /* fail_idx: chunk the one-thread decoder takes over from, or -1 */
		bool new_member, member_end; uint32_t crc_expect, isize; bool stream_end;      /* stream_end: nothing follows this wave */
		wave_t() : A(0), fail_idx(-1), new_member(false), member_end(false), crc_expect(0), isize(0), stream_end(false) {}
	};
	/* the whole file; false on error (err).  The calling thread reads, searches and decodes wave after wave (T threads); a second thread
	 * stitches the waves behind it (markers -> bytes and CRC by T threads, bytes to the sink): a wave's decode needs only the bit where
	 * the previous one ended, not its window */
	bool run(const sink_t &sink)
	{
		struct stat sb; if (fstat(fd, &sb) != 0) { err = "cannot stat the input"; return false; }
		file_size = (uint64_t)sb.st_size;
		std::mutex mu; std::condition_variable cv; std::vector<std::unique_ptr<wave_t> > ready, spare; bool closed = false; std::atomic<bool> stop(false);
		const char *stitch_err = 0; bool stitch_ok = true;
		for (int k = 0; k < 3; ++k) spare.emplace_back(new wave_t());
		/* the stitcher's helpers run beside the next wave's decoders: as many as those where the host has the cores, half otherwise */
		const int R = std::thread::hardware_concurrency() >= (unsigned)(3 * T) ? T : std::max(1, T / 2);
		std::thread stitcher([&]() {
			std::vector<uint8_t> window, bytes; uint64_t member_out = 0; uLong member_crc = crc32(0L, Z_NULL, 0);
			std::vector<uLong> part_crc((size_t)R); std::vector<size_t> part_len((size_t)R);
			for (;;) {
				std::unique_ptr<wave_t> W;
				{ std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return !ready.empty() || closed; }); if (ready.empty()) break; W = std::move(ready.front()); ready.erase(ready.begin()); }
				if (stitch_ok) {
					if (W->new_member) { window.clear(); member_out = 0; member_crc = crc32(0L, Z_NULL, 0); }
					for (size_t ci = 0; ci < W->chain.size() && stitch_ok; ++ci) {
						chunk_t &c = W->ch[(size_t)W->chain[ci]];
						if (W->chain[ci] == W->fail_idx) {   /* the rest of the member (and of the file) through the one-thread decoder, from the bit where the last good piece ended */
							if (!serial_rest(sink, W->A * 8 + c.start, window, member_out, member_crc)) { stitch_err = err; stitch_ok = false; }
							break;
						}
						bytes.resize(c.n);
						const uint16_t *src = c.out.data() + WIN; const uint8_t *wnd = window.data(); const size_t wn = window.size();
						std::atomic<bool> bad(false);
						parallel(R, [&](int t) {
							const size_t a = c.n * (size_t)t / (size_t)R, b = c.n * (size_t)(t + 1) / (size_t)R;
							part_len[(size_t)t] = b - a;
							for (size_t i0 = a; i0 < b; i0 += 1024) {   /* blocks of 1024: a plain narrowing loop (vectorised); only blocks that hold a marker are done again */
								const size_t i1 = std::min(b, i0 + 1024); uint16_t any = 0;
								for (size_t i = i0; i < i1; ++i) { const uint16_t v = src[i]; any |= v; bytes[i] = (uint8_t)v; }
								if (!(any & 0x8000u)) continue;
								for (size_t i = i0; i < i1; ++i) {
									const uint16_t v = src[i];
									if (v & 0x8000u) { const size_t m = v & 0x7fffu; if (m + wn < (size_t)WIN) { bad = true; bytes[i] = 0; } else bytes[i] = wnd[m - ((size_t)WIN - wn)]; }
								}
							}
							part_crc[(size_t)t] = b > a ? crc32(crc32(0L, Z_NULL, 0), bytes.data() + a, (uInt)(b - a)) : crc32(0L, Z_NULL, 0);
						});
						if (bad) { stitch_err = "distance too far back"; stitch_ok = false; break; }
						for (int t = 0; t < R; ++t) if (part_len[(size_t)t]) member_crc = crc32_combine(member_crc, part_crc[(size_t)t], (z_off_t)part_len[(size_t)t]);
						member_out += c.n;
						if (c.n >= (size_t)WIN) window.assign(bytes.end() - WIN, bytes.end());
						else { window.insert(window.end(), bytes.begin(), bytes.end()); if (window.size() > (size_t)WIN) window.erase(window.begin(), window.end() - WIN); }
						const bool closes = c.final && W->member_end;
						if (closes) {
							if ((uint32_t)member_out != W->isize) { stitch_err = "length in the gzip trailer does not match"; stitch_ok = false; break; }
							if ((uint32_t)member_crc != W->crc_expect) { stitch_err = "CRC-32 in the gzip trailer does not match"; stitch_ok = false; break; }
						}
						size_t o2 = 0;                                          /* the bytes go on in pieces of <= 4 MB; the last piece of a member says so */
						do {
							const size_t k = std::min<size_t>(c.n - o2, (size_t)4 << 20); const bool lastp = o2 + k == c.n;
							if (!sink(bytes.data() + o2, k, lastp && closes, W->crc_expect)) { stitch_err = "cancelled"; stitch_ok = false; break; }
							o2 += k;
						} while (o2 < c.n);
					}
					if (!stitch_ok) stop = true;
				}
				{ std::lock_guard<std::mutex> l(mu); spare.push_back(std::move(W)); }
				cv.notify_all();
			}
		});
		auto finish = [&](bool ok, const char *e) -> bool {
			{ std::lock_guard<std::mutex> l(mu); closed = true; }
			cv.notify_all(); stitcher.join();
			if (!ok) { err = e; return false; }
			if (!stitch_ok) { err = stitch_err ? stitch_err : "damaged deflate stream"; return false; }
			return true;
		};
		uint64_t member_byte = 0;                                          /* file offset of the member being decoded */
		const size_t SLACK = (size_t)4 << 20;
		std::vector<uint8_t> wbuf;
		bool first_member = true;
		while (member_byte < file_size && !stop) {
			/* ---- member header (one thread, through the ordinary decoder's parser) ---- */
			uint64_t pos;                                                  /* bit position in the FILE of the next block */
			{
				/* the header's fields have no length limit (FNAME / FCOMMENT run to their NUL): a header that does not end inside the bytes
				 * read so far is read again with more of the file, and is truncated only when the file itself ends inside it */
				size_t hdr_len = 0; bool hdr_done = false, no_member = false;
				for (size_t want = 70000; !hdr_done; want <<= 3) {
					const size_t n = (size_t)std::min<uint64_t>(file_size - member_byte, want);
					std::vector<uint8_t> hb(n + 16, 0);
					if (!pread_all(fd, hb.data(), n, member_byte)) return finish(false, "read error");
					fast_gz_t g(-1); g.ib = hb.data(); g.ireal = n; g.iend = n + 16; g.eof_in = true; g.ip = 0; g.member_done = !first_member;
					if (g.header()) { hdr_done = true; no_member = g.st == fast_gz_t::S_DONE; hdr_len = g.ip; }
					else if (n == file_size - member_byte || !g.err || strcmp(g.err, "truncated gzip header")) return finish(false, g.err ? g.err : "damaged gzip header");
				}
				if (no_member) return finish(true, 0);                        /* bytes that are no member after a complete one */
				pos = (member_byte + hdr_len) * 8;
			}
			first_member = false;
			bool member_end = false, new_member = true;
			while (!member_end && !stop) {
				/* ---- one wave: read, find block starts, decode ---- */
				std::unique_ptr<wave_t> W;
				{ std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return !spare.empty() || stop.load(); }); if (spare.empty()) break; W = std::move(spare.back()); spare.pop_back(); }
				const uint64_t A = pos >> 3;                                 /* file offset of wbuf[0] */
				const size_t want = (size_t)std::min<uint64_t>(file_size - A, (uint64_t)T * C + SLACK);
				if (wbuf.size() < want + 64) wbuf.resize(want + 64);
				if (!pread_all(fd, wbuf.data(), want, A)) return finish(false, "read error");
				memset(wbuf.data() + want, 0, 64);
				const size_t real = want, padded = want + 64;
				const uint64_t soft_end = std::min<uint64_t>((uint64_t)T * C, want) * 8;
				std::vector<chunk_t> &ch = W->ch;
				if (ch.size() != (size_t)T) ch.resize((size_t)T);
				for (chunk_t &c : ch) { c.start = c.end = 0; c.valid = c.final = c.failed = false; c.n = 0; }
				W->A = A; W->chain.clear(); W->fail_idx = -1; W->new_member = new_member; W->member_end = false; W->stream_end = false; new_member = false;
				ch[0].start = pos - A * 8; ch[0].valid = true;
				parallel(T, [&](int k) {
					if (k == 0 || (uint64_t)k * C >= want) return;
					const int64_t p = find_block(wbuf.data(), real, padded, (uint64_t)k * C * 8, std::min<uint64_t>((uint64_t)(k + 1) * C, want) * 8);
					if (p >= 0) { ch[(size_t)k].start = (uint64_t)p; ch[(size_t)k].valid = true; }
				});
				std::vector<uint64_t> stops;
				for (int k = 1; k < T; ++k) if (ch[(size_t)k].valid) stops.push_back(ch[(size_t)k].start);
				const bool to_eof = A + want == file_size;
				parallel(T, [&](int k) { if (ch[(size_t)k].valid) decode_chunk(ch[(size_t)k], wbuf.data(), real, padded, stops, soft_end, to_eof); });
				/* ---- the chunks that abut, in stream order (no window needed for this) ---- */
				bool handed_over = false;
				for (size_t cur = 0;;) {
					chunk_t &c = ch[cur];
					W->chain.push_back((int)cur);
					if (c.failed) { W->fail_idx = (int)cur; handed_over = true; break; }
					pos = A * 8 + c.end;
					if (c.final) {   /* trailer: CRC-32 and length of the member, byte-aligned after the final block */
						const uint64_t tb = (pos + 7) >> 3; uint8_t tr[8];
						if (tb + 8 > file_size || !pread_all(fd, tr, 8, tb)) return finish(false, "truncated gzip trailer");
						memcpy(&W->crc_expect, tr, 4); memcpy(&W->isize, tr + 4, 4);
						W->member_end = true; member_end = true; member_byte = tb + 8;
						break;
					}
					size_t nxt = cur + 1; while (nxt < (size_t)T && !(ch[nxt].valid && ch[nxt].start == c.end)) ++nxt;
					if (nxt >= (size_t)T) break;                               /* the next wave starts at pos */
					cur = nxt;
				}
				{ std::lock_guard<std::mutex> l(mu); ready.push_back(std::move(W)); }
				cv.notify_all();
				if (handed_over) return finish(true, 0);                      /* the stitcher's one-thread decoder finishes the file */
			}
		}
		return finish(true, 0);
	}

	/* rest of the current member (and whatever follows it) by the one-thread decoder: entered at file bit `bit` with `window` before it */
	bool serial_rest(const sink_t &sink, uint64_t bit, const std::vector<uint8_t> &window, uint64_t member_out, uLong crc)
	{
		if (lseek(fd, (off_t)(bit >> 3), SEEK_SET) < 0) { err = "cannot seek"; return false; }
		fast_gz_t g(fd);
		if (!g.fill_input()) { err = g.err; return false; }
		g.refill(); (void)g.bits((int)(bit & 7));
		g.obuf.resize((size_t)WIN + 1024);
		if (!window.empty()) memcpy(g.obuf.data() + WIN - window.size(), window.data(), window.size());
		g.member_out = member_out; g.st = fast_gz_t::S_BLOCK;   /* window.size() = min(member_out, 32 KB): what a distance may reach */
		for (;;) {
			const uint8_t *p; bool mend = false;
			const long n = g.read_chunk(&p, (size_t)4 << 20, &mend);
			if (n < 0) { err = g.err ? g.err : "damaged deflate stream"; return false; }
			if (!n && !mend) return true;
			if (n) crc = crc32(crc, p, (uInt)n);
			if (mend) { if ((uint32_t)crc != g.crc_expect) { err = "CRC-32 in the gzip trailer does not match"; return false; } crc = crc32(0L, Z_NULL, 0); }
			if (!sink(p, (size_t)n, mend, g.crc_expect)) { err = "cancelled"; return false; }
		}
	}
};
#endif
