/*
 * fastq.h -- FASTQ / FASTA ingest for `bwa mem`: the record grammar of klib's kseq_read
 * (/root/reference/src/samtools-1.3.1/htslib-1.3.1/htslib/kseq.h:189-229 -- header char '>' or '@', name up to the
 * first blank, the rest of the line as comment, sequence lines until a line starting with '>', '+' or '@',
 * quality lines until as many characters as bases; -2 on a truncated quality string) followed by upstream bseq.c's
 * trim_readno ("/1", "/2" suffixes dropped).  One reader thread per input file inflates (zlib) and parses into
 * blocks of records held in flat arenas -- no per-read allocation -- and hands them to the batch assembler through a
 * bounded channel, so inflate + parse of both files overlap each other and the GPU.
 */
#ifndef SSG_FASTQ_H
#define SSG_FASTQ_H
#include <stdint.h>
#include <string.h>
#include <ctype.h>
#include <zlib.h>
#include <vector>
#include <deque>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <functional>
#include <memory>
#include <string>
#include <atomic>
#include <algorithm>
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>
#include <sys/stat.h>
#include "fast_inflate.h"
#include "fast_inflate_mt.h"

template <class T> struct chan_t {      /* bounded single-producer / single-consumer channel */
	std::mutex mu; std::condition_variable cv; std::deque<T> q; size_t cap; bool closed, dead;
	explicit chan_t(size_t cap_ = 2) : cap(cap_), closed(false), dead(false) {}
	void push(T v) { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return q.size() < cap || dead; }); if (dead) return; q.push_back(std::move(v)); cv.notify_all(); }
	void abandon() { std::lock_guard<std::mutex> l(mu); dead = true; q.clear(); cv.notify_all(); }   /* the consumer is gone: producers must not wait for it */
	bool is_dead() { std::lock_guard<std::mutex> l(mu); return dead; }
	bool pop(T &v) { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; v = std::move(q.front()); q.pop_front(); cv.notify_all(); return true; }
	void close() { std::lock_guard<std::mutex> l(mu); closed = true; cv.notify_all(); }
};

struct fq_block_t {                     /* a run of consecutive records of one file */
	std::vector<char> txt;              /* names and comments, NUL-terminated */
	std::vector<uint32_t> name_o, com_o;    /* offsets into txt; com_o = UINT32_MAX: no comment */
	std::vector<uint8_t> seq;           /* nt4 codes, concatenated */
	std::vector<char> qual;             /* quality strings, each NUL-terminated, only for records with has_q */
	std::vector<uint32_t> seq_o, qual_o;    /* n + 1 / n offsets */
	std::vector<uint8_t> has_q;
	int n, err;                         /* err: 0, or -2 truncated / malformed */
	fq_block_t() : n(0), err(0) { seq_o.push_back(0); }
};

struct fq_stream_t {                    /* kstream over gzread; for compressed input the inflate runs in a thread of its own, a few 4-MB chunks ahead of the parser */
	typedef std::vector<unsigned char> chunk_t;
	gzFile fp; chunk_t buf; int begin, end; bool is_eof;
	chan_t<std::unique_ptr<chunk_t> > full, empty; std::thread th;
	std::atomic<bool> stop{false};
	std::atomic<int> io_err{0};         /* the compressed stream was damaged or cut short: the parser reports the file as malformed at its end */
	bool threaded;                      /* compressed input only: for a plain file the hand-over costs more than the read */
	bool bgzf;                          /* blocked gzip (bgzip, htslib bgzf.c:298-342): independent <= 64 KB members with their size in the header, inflated by several threads */
	int rfd; size_t roff, rend;         /* ranged mode (fq_feed_t's parallel parse of a plain file): bytes [roff, rend) of rfd through pread */
	const unsigned char *mp; size_t mn, mo; fq_stream_t *chain;   /* memory mode (rfd = -2): bytes mp[0..mn), then -- if chain -- the decoded chunks of another stream (fq_feed_t::parse_stream) */
	/* BGZF member at p (n bytes available): its total size, or 0 when p does not start one */
	static size_t bgzf_member(const unsigned char *p, size_t n)
	{
		if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
		const size_t xlen = p[10] | (size_t)p[11] << 8;
		if (n < 12 + xlen) return 0;
		for (size_t o = 12; o + 4 <= 12 + xlen; ) { const size_t sl = p[o + 2] | (size_t)p[o + 3] << 8; if (p[o] == 'B' && p[o + 1] == 'C' && sl == 2) return (size_t)(p[o + 4] | (size_t)p[o + 5] << 8) + 1; o += 4 + sl; }
		return 0;
	}
	void fast_gz_loop(int fd)
	{	/* inflate (fast_inflate.h) -> CRC-32 of every member against its trailer -> the parser: three threads in a row, 4 MB chunks */
		struct piece_t { std::unique_ptr<chunk_t> c; bool member_end; uint32_t crc; bool check; };   /* check: the CRC is this pipeline's to verify (one decoding thread); several threads verify it themselves */
		chan_t<piece_t> mid(4);
		std::thread t_crc([this, &mid]() {
			uLong crc = crc32(0L, Z_NULL, 0); piece_t p;
			while (mid.pop(p)) {
				if (io_err.load()) continue;
				if (p.check && p.c && !p.c->empty()) crc = crc32(crc, p.c->data(), (uInt)p.c->size());
				if (p.check && p.member_end) { if ((uint32_t)crc != p.crc) { io_err = 1; continue; } crc = crc32(0L, Z_NULL, 0); }
				if (p.c && !p.c->empty()) full.push(std::move(p.c));
			}
			full.close();
		});
		bool check_crc = true;
		auto hand_on = [&](const uint8_t *d, size_t n, bool mend, uint32_t crc_expect) -> bool {
			piece_t p; p.member_end = mend; p.crc = crc_expect; p.check = check_crc;
			{ std::unique_lock<std::mutex> l(empty.mu); if (!empty.q.empty()) { p.c = std::move(empty.q.front()); empty.q.pop_front(); } }
			if (!p.c) p.c.reset(new chunk_t());
			p.c->assign(d, d + n);
			mid.push(std::move(p));
			return !stop.load() && !io_err.load();
		};
		/* several decoding threads for the one stream (fast_inflate_mt.h) where the host has cores to spare: SSG_GZ_THREADS, default 12 on
		 * hosts with 64 hardware threads or more, 8 from 32, otherwise one */
		const unsigned hw = std::thread::hardware_concurrency();
		int gt = hw >= 64 ? 12 : hw >= 32 ? 8 : 1; { const char *e = getenv("SSG_GZ_THREADS"); if (e && atoi(e) > 0) gt = atoi(e); }
		if (gt > 1) {
			check_crc = false;
			size_t gc = (size_t)2 << 20; { const char *e = getenv("SSG_GZ_CHUNK"); if (e && atol(e) > 0) gc = (size_t)atol(e); }   /* compressed bytes per thread and wave (tests make it small) */
			fast_gz_mt_t g(fd, gt, gc);
			if (!g.run(hand_on) && !stop.load()) io_err = 1;
		} else {
			fast_gz_t g(fd);
			while (!stop.load() && !io_err.load()) {
				const uint8_t *d; bool mend = false;
				const long n = g.read_chunk(&d, (size_t)4 << 20, &mend);
				if (n < 0) { io_err = 1; break; }
				if (!n && !mend) break;
				if (!hand_on(d, (size_t)n, mend, g.crc_expect)) break;
			}
		}
		mid.close();
		t_crc.join();
	}
	void bgzf_loop(int fd, int n_threads)
	{	/* batches of members: located by their headers, inflated in parallel, handed to the parser in file order */
		std::vector<unsigned char> raw; size_t pos = 0; bool eof = false;
		while (!stop.load()) {
			if (pos > ((size_t)8 << 20)) { raw.erase(raw.begin(), raw.begin() + (long)pos); pos = 0; }
			while (!eof && raw.size() - pos < ((size_t)24 << 20)) {
				const size_t old = raw.size(); raw.resize(old + ((size_t)8 << 20));
				ssize_t r = read(fd, raw.data() + old, (size_t)8 << 20);
				if (r < 0) { if (errno == EINTR) { raw.resize(old); continue; } io_err = 1; r = 0; }   /* a read error is not the end of the file */
				raw.resize(old + (size_t)r);
				if (r == 0) eof = true;
			}
			struct span_t { size_t off, len, isz, out; };
			std::vector<span_t> sp; size_t total = 0, q = pos;
			while (sp.size() < 512) {
				const size_t m = bgzf_member(raw.data() + q, raw.size() - q);
				if (!m || q + m > raw.size()) break;
				span_t x; x.off = q; x.len = m; x.isz = raw[q + m - 4] | (size_t)raw[q + m - 3] << 8 | (size_t)raw[q + m - 2] << 16 | (size_t)raw[q + m - 1] << 24; x.out = total;
				total += x.isz; q += m; sp.push_back(x);
			}
			if (sp.empty()) {
				/* end of file -- unless bytes are left that are (the beginning of) a gzip member: a BGZF member cut short, or a member
				 * without the BGZF field, which this loop cannot hand on.  Either way the input is not what was read so far: the run fails
				 * (zlib reports both; bytes that are no gzip member at all are ignored, as gzread ignores them) */
				if (raw.size() - pos >= 2 && raw[pos] == 0x1f && raw[pos + 1] == 0x8b) io_err = 1;
				else if (raw.size() - pos == 1 && raw[pos] == 0x1f) io_err = 1;
				break;
			}
			pos = q;
			std::unique_ptr<chunk_t> c(new chunk_t(total));
			std::atomic<size_t> next(0); std::atomic<int> bad(0);
			auto work = [&]() {
				for (;;) {
					const size_t k = next.fetch_add(1);
					if (k >= sp.size()) break;
					const unsigned char *h = raw.data() + sp[k].off; const size_t xlen = h[10] | (size_t)h[11] << 8;
					if (sp[k].len < 12 + xlen + 8) { bad = 1; continue; }
					z_stream zs; memset(&zs, 0, sizeof(zs));
					zs.next_in = (Bytef*)(h + 12 + xlen); zs.avail_in = (uInt)(sp[k].len - 12 - xlen - 8); zs.next_out = c->data() + sp[k].out; zs.avail_out = (uInt)sp[k].isz;
					unsigned char none = 0; if (!sp[k].isz) zs.next_out = &none;   /* an empty member (the end-of-file block) still has a deflate stream and a CRC */
					if (inflateInit2(&zs, -15) != Z_OK || inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.total_out != sp[k].isz) bad = 1;
					inflateEnd(&zs);
					const unsigned char *tr = h + sp[k].len - 8;   /* the member's own CRC-32 of its output */
					const uint32_t want = tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24;
					if (!bad && (uint32_t)crc32(crc32(0L, Z_NULL, 0), c->data() + sp[k].out, (uInt)sp[k].isz) != want) bad = 1;
				}
			};
			std::vector<std::thread> w;
			for (int t = 1; t < n_threads; ++t) w.emplace_back(work);
			work();
			for (auto &x : w) x.join();
			if (bad) { io_err = 1; break; }                      /* the parser reports the file as malformed at its end */
			if (total) full.push(std::move(c));
		}
		full.close();
	}
	fq_stream_t(int fd, size_t off, size_t lim) : fp(0), begin(0), end(0), is_eof(false), full(1), empty(1), threaded(false), bgzf(false), rfd(fd), roff(off), rend(lim), mp(0), mn(0), mo(0), chain(0) { buf.resize((size_t)1 << 20); }
	fq_stream_t(const unsigned char *p, size_t n, fq_stream_t *then) : fp(0), begin(0), end(0), is_eof(false), full(1), empty(1), threaded(false), bgzf(false), rfd(-2), roff(0), rend(0), mp(p), mn(n), mo(0), chain(then) { buf.resize((size_t)1 << 20); }
	bool had_io_err() const { return io_err.load() != 0 || (chain && chain->io_err.load() != 0); }
	explicit fq_stream_t(gzFile f, const char *path = 0) : fp(f), begin(0), end(0), is_eof(false), full(4), empty(8), bgzf(false), rfd(-1), roff(0), rend(0), mp(0), mn(0), mo(0), chain(0)
	{
		gzbuffer(fp, 1 << 20);
		threaded = !gzdirect(fp);
		if (!threaded) { buf.resize((size_t)1 << 20); return; }
		if (path) {
			struct stat sb;
			const int fd = (stat(path, &sb) == 0 && S_ISREG(sb.st_mode)) ? open(path, O_RDONLY) : -1;   /* a pipe must not lose bytes to this look */
			unsigned char h[18] = { 0 };
			const ssize_t nh = fd >= 0 ? read(fd, h, 18) : -1;
			if (nh == 18 && bgzf_member(h, 18) && lseek(fd, 0, SEEK_SET) == 0) {
				bgzf = true;
				int nt = 8; { const char *e = getenv("SSG_BGZF_THREADS"); if (e && atoi(e) > 0) nt = atoi(e); }
				th = std::thread([this, fd, nt]() { bgzf_loop(fd, nt); close(fd); });
				return;
			}
			const char *gf = getenv("SSG_GZ_FAST");
			if (nh >= 2 && h[0] == 0x1f && h[1] == 0x8b && !(gf && !strcmp(gf, "0")) && lseek(fd, 0, SEEK_SET) == 0) {
				th = std::thread([this, fd]() { fast_gz_loop(fd); close(fd); });   /* plain gzip in a regular file: this repository's decoder, CRC on a thread of its own */
				return;
			}
			if (fd >= 0) close(fd);
		}
		th = std::thread([this]() {
			while (!stop.load()) {
				std::unique_ptr<chunk_t> c;
				{ std::unique_lock<std::mutex> l(empty.mu); if (!empty.q.empty()) { c = std::move(empty.q.front()); empty.q.pop_front(); } }   /* a recycled chunk if one is back */
				if (!c) c.reset(new chunk_t());
				c->resize((size_t)4 << 20);
				const int n = gzread(fp, c->data(), (unsigned)c->size());
				if (n < 0) io_err = 1;
				if (n <= 0) break;
				c->resize((size_t)n);
				full.push(std::move(c));
			}
			full.close();
		});
	}
	/* provider mode (rank mode, bwa_main.cpp): the bytes come from a function that fills a chunk at a time (false = the end) -- the byte ranges of
	 * a plain file that hold this rank's batches; a thread of its own calls it, the parser takes the chunks as it takes a decoder's */
	typedef std::function<bool(chunk_t&)> provider_t;
	explicit fq_stream_t(provider_t prov) : fp(0), begin(0), end(0), is_eof(false), full(4), empty(8), threaded(true), bgzf(false), rfd(-1), roff(0), rend(0), mp(0), mn(0), mo(0), chain(0)
	{
		th = std::thread([this, prov]() {
			while (!stop.load()) {
				std::unique_ptr<chunk_t> c;
				{ std::unique_lock<std::mutex> l(empty.mu); if (!empty.q.empty()) { c = std::move(empty.q.front()); empty.q.pop_front(); } }
				if (!c) c.reset(new chunk_t());
				c->clear();
				if (!prov(*c)) break;
				if (!c->empty()) full.push(std::move(c));
			}
			full.close();
		});
	}
	~fq_stream_t() { stop.store(true); if (threaded) { std::unique_ptr<chunk_t> c; while (full.pop(c)) {} } if (th.joinable()) th.join(); }
	inline bool fill()
	{
		if (is_eof) return false;
		if (rfd == -2) {
			if (mo < mn) { const size_t k = std::min(buf.size(), mn - mo); memcpy(buf.data(), mp + mo, k); mo += k; begin = 0; end = (int)k; return true; }
			if (chain && !chain->threaded) {                        /* an uncompressed stream: its own read, its buffer taken over */
				if (!chain->fill()) { begin = end = 0; is_eof = true; return false; }
				buf.swap(chain->buf); begin = chain->begin; end = chain->end; chain->begin = chain->end = 0;
				if (chain->buf.size() < ((size_t)1 << 20)) chain->buf.resize((size_t)1 << 20);
				return true;
			}
			std::unique_ptr<chunk_t> c;
			if (!chain || !chain->full.pop(c)) { begin = end = 0; is_eof = true; return false; }
			buf.swap(*c);
			{ std::lock_guard<std::mutex> l(chain->empty.mu); if (chain->empty.q.size() < 8) chain->empty.q.push_back(std::move(c)); }
			begin = 0; end = (int)buf.size();
			return true;
		}
		if (rfd >= 0) {
			begin = end = 0;
			while (roff < rend) {
				const ssize_t r = pread(rfd, buf.data(), std::min(buf.size(), rend - roff), (off_t)roff);
				if (r < 0 && errno == EINTR) continue;
				if (r > 0) { end = (int)r; roff += (size_t)r; }
				break;
			}
			if (end <= 0) { is_eof = true; return false; }
			return true;
		}
		if (!threaded) { begin = 0; end = gzread(fp, buf.data(), (unsigned)buf.size()); if (end <= 0) { end = 0; is_eof = true; return false; } return true; }
		std::unique_ptr<chunk_t> c;
		if (!full.pop(c)) { begin = end = 0; is_eof = true; return false; }
		buf.swap(*c);
		{ std::lock_guard<std::mutex> l(empty.mu); if (empty.q.size() < 8) empty.q.push_back(std::move(c)); }
		begin = 0; end = (int)buf.size();
		return true;
	}
	inline int getc() { if (begin >= end && !fill()) return -1; return buf[begin++]; }
	/* ks_getuntil2: append to out until the delimiter (0 = any blank, 2 = end of line); returns -1 at EOF with nothing read; *dret = delimiter */
	/* field0: where the string this call appends to began (kseq strips a trailing CR when the WHOLE string is longer than one character,
	 * ks_getuntil2's `str->l > 1`: a sequence or quality line continues a string the caller already started) */
	template <class V> int getuntil(int delim, V &out, int *dret, size_t field0 = (size_t)-1)
	{
		bool got = false; size_t l0 = out.size();
		if (field0 == (size_t)-1) field0 = l0;
		if (dret) *dret = 0;
		for (;;) {
			if (begin >= end) { if (!fill()) break; }
			int i;
			if (delim == 2) { const unsigned char *p = (const unsigned char*)memchr(buf.data() + begin, '\n', (size_t)(end - begin)); i = p ? (int)(p - buf.data()) : end; }
			else { for (i = begin; i < end; ++i) { const unsigned c = buf[i]; if (c == ' ' || (c - 9u) <= 4u) break; } }   /* isspace in the C locale, inline */
			got = true;
			out.insert(out.end(), buf.data() + begin, buf.data() + i);
			begin = i + 1;
			if (i < end) { if (dret) *dret = buf[i]; break; }
		}
		if (!got && is_eof && begin >= end) return -1;
		if (delim == 2 && out.size() - field0 > 1 && out.back() == '\r') out.pop_back();
		return out.size() > l0 ? (int)(out.size() - l0) : 0;
	}
};

struct fq_nt4_tab_t { uint8_t t[256]; fq_nt4_tab_t() { memset(t, 4, 256); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; } };
static inline const uint8_t *fq_nt4_tab() { static const fq_nt4_tab_t T; return T.t; }
static inline uint8_t fq_nt4(int c) { return fq_nt4_tab()[(unsigned char)c]; }

struct fq_reader_t {
	fq_stream_t ks; int last_char; bool keep_comment;
	bool qual_eof;                      /* the quality loop ran into the end of the stream: only the end of the FILE may do that */
	fq_reader_t(gzFile f, bool kc, const char *path = 0) : ks(f, path), last_char(0), keep_comment(kc), qual_eof(false) {}
	fq_reader_t(int fd, size_t off, size_t lim, bool kc) : ks(fd, off, lim), last_char(0), keep_comment(kc), qual_eof(false) {}
	fq_reader_t(const unsigned char *p, size_t n, fq_stream_t *then, bool kc) : ks(p, n, then), last_char(0), keep_comment(kc), qual_eof(false) {}
	/* kseq_read + kseq2bseq1 + trim_readno into block b; >= 0 length, -1 EOF, -2 truncated quality */
	int next(fq_block_t &b)
	{
		int c;
		if (last_char == 0) { while ((c = ks.getc()) != -1 && c != '>' && c != '@') {} if (c == -1) return ks.had_io_err() ? -2 : -1; last_char = c; }   /* a damaged compressed stream ends early: that is an error, not the end of the reads */
		const size_t name0 = b.txt.size();
		if (ks.getuntil(0, b.txt, &c) < 0) return ks.had_io_err() ? -2 : -1;
		size_t name_end = b.txt.size();
		if (name_end - name0 > 2 && b.txt[name_end - 2] == '/' && isdigit((unsigned char)b.txt[name_end - 1])) { b.txt.resize(name_end - 2); name_end -= 2; }   /* trim_readno */
		b.txt.push_back(0);
		uint32_t com = UINT32_MAX;
		if (c != '\n') {
			const size_t c0 = b.txt.size();
			ks.getuntil(2, b.txt, 0);
			if (keep_comment && b.txt.size() > c0) { b.txt.push_back(0); com = (uint32_t)c0; } else b.txt.resize(c0);
		}
		const size_t s0 = b.seq.size();
		while ((c = ks.getc()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;
			b.seq.push_back((uint8_t)c);
			ks.getuntil(2, b.seq, 0, s0);
		}
		if (c == '>' || c == '@') last_char = c;
		const size_t l_seq = b.seq.size() - s0;
		bool hq = false; const size_t q0 = b.qual.size();
		if (c == '+') {
			while ((c = ks.getc()) != -1 && c != '\n') {}
			if (c == -1) return -2;
			int r;
			while ((r = ks.getuntil(2, b.qual, 0, q0)) >= 0 && b.qual.size() - q0 < l_seq) {}
			if (r < 0) qual_eof = true;
			last_char = 0;
			if (b.qual.size() - q0 != l_seq) return -2;
			b.qual.push_back(0); hq = true;
		}
		{ const uint8_t *const T = fq_nt4_tab(); uint8_t *const q = b.seq.data(); const size_t e = b.seq.size(); for (size_t i = s0; i < e; ++i) q[i] = T[q[i]]; }
		b.name_o.push_back((uint32_t)name0); b.com_o.push_back(com); b.seq_o.push_back((uint32_t)b.seq.size());
		b.qual_o.push_back((uint32_t)q0); b.has_q.push_back(hq ? 1 : 0); ++b.n;
		return (int)l_seq;
	}
};

/* reader thread: blocks of up to `per_block` records into the channel; the last block carries err / a short count, then the channel closes */
struct fq_block_pool_t {                /* blocks go back to the reader when their last user lets go: warm memory instead of fresh pages per block */
	std::mutex mu; std::vector<fq_block_t*> free_;
	~fq_block_pool_t() { for (fq_block_t *b : free_) delete b; }
	fq_block_t *get()
	{
		fq_block_t *b = 0;
		{ std::lock_guard<std::mutex> l(mu); if (!free_.empty()) { b = free_.back(); free_.pop_back(); } }
		if (!b) return new fq_block_t();
		b->txt.clear(); b->name_o.clear(); b->com_o.clear(); b->seq.clear(); b->qual.clear(); b->seq_o.clear(); b->qual_o.clear(); b->has_q.clear();
		b->seq_o.push_back(0); b->n = 0; b->err = 0;
		return b;
	}
	void put(fq_block_t *b) { std::lock_guard<std::mutex> l(mu); if (free_.size() < 4096) { free_.push_back(b); return; } delete b; }
};
struct fq_feed_t {
	chan_t<std::shared_ptr<fq_block_t> > ch; std::thread th;   /* shared: a batch keeps the blocks its names and qualities point into */
	std::shared_ptr<fq_block_pool_t> pool;
	typedef std::shared_ptr<fq_block_t> blk_t;
	blk_t fresh(int per_block)
	{
		std::shared_ptr<fq_block_pool_t> pl = pool;
		blk_t b(pl->get(), [pl](fq_block_t *x) { pl->put(x); });
		b->txt.reserve((size_t)per_block * 48); b->seq.reserve((size_t)per_block * 160); b->qual.reserve((size_t)per_block * 160);
		return b;
	}
	/* everything `rd` holds, as blocks of up to per_block records through `sink`; true when the stream ended in the middle of a record */
	template <class F> bool drain(fq_reader_t &rd, int per_block, F sink)
	{
		for (;;) {
			blk_t b = fresh(per_block);
			int rc = 0;
			while (b->n < per_block && (rc = rd.next(*b)) >= 0) {}
			if (rc == -2) b->err = -2;
			const bool last = rc < 0;
			if (b->n || b->err) sink(std::move(b));
			if (last) return rc == -2 || rd.qual_eof;
			if (ch.is_dead()) return false;
		}
	}
	/* A record may start at q (a line start): '@' line, a line that does not start with '@' '+' '>', a '+' line, a line as long as the
	 * second one.  Only a hint -- parse_plain() proves every cut it uses. */
	static bool looks_like_record(const unsigned char *p, size_t n, size_t q)
	{
		if (q >= n || p[q] != '@') return false;
		size_t l[5]; l[0] = q;
		for (int k = 1; k < 5; ++k) { const unsigned char *e = (const unsigned char*)memchr(p + l[k - 1], '\n', n - l[k - 1]); if (!e) return false; l[k] = (size_t)(e - p) + 1; }
		if (l[2] - l[1] < 2 || p[l[1]] == '@' || p[l[1]] == '+' || p[l[1]] == '>' || p[l[2]] != '+') return false;
		return l[4] - l[3] == l[2] - l[1] && (l[4] >= n || p[l[4]] == '@');
	}
	/* A plain (not compressed) regular file, parsed by several threads: the file is cut at lines that look like record starts, each
	 * piece goes through the same kseq grammar as the serial reader (fq_reader_t over a byte range) and the pieces are handed on in
	 * file order.  What makes this exact for ANY input, wrapped or malformed ones included: a piece that was entered at a true
	 * record start and does not run out of bytes inside a quality string leaves the parser where the serial one would be (between
	 * records, or after a sequence without quality -- and the next piece starts with '@' at a line start, which ends such a
	 * sequence in the serial parser too).  The first piece starts at byte 0, so by induction every cut is a true record start as long
	 * as no earlier piece ended inside a quality string; the first piece that does (a '@' quality line was taken for a header, or
	 * the file really is truncated there) is thrown away and the rest of the file is parsed serially from that piece's start. */
	bool parse_plain(const std::string &path, bool keep_comment, int per_block, int threads_hint)
	{
		size_t piece = (size_t)48 << 20; int T = threads_hint > 0 ? threads_hint : std::thread::hardware_concurrency() >= 32 ? 6 : 4;
		{ const char *e = getenv("SSG_FASTQ_PIECE"); if (e && atol(e) > 0) piece = (size_t)atol(e); }
		{ const char *e = getenv("SSG_FASTQ_THREADS"); if (e) T = atoi(e); }
		if (T < 2) return false;
		struct stat sb;
		if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode) || (size_t)sb.st_size < 2 * piece) return false;
		const int fd = open(path.c_str(), O_RDONLY);
		if (fd < 0) return false;
		unsigned char magic[2];
		if (pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) { close(fd); return false; }   /* gzip: the inflating readers */
		const size_t size = (size_t)sb.st_size;
		std::vector<size_t> cut(1, 0);
		{
			std::vector<unsigned char> w((size_t)256 << 10);
			for (size_t pos = piece; pos < size; pos += piece) {
				const ssize_t r = pread(fd, w.data(), w.size(), (off_t)pos);
				if (r <= 0) break;
				const size_t n = (size_t)r;
				const unsigned char *e = (const unsigned char*)memchr(w.data(), '\n', n);
				while (e) {
					const size_t q = (size_t)(e - w.data()) + 1;
					if (looks_like_record(w.data(), n, q)) { if (pos + q > cut.back()) cut.push_back(pos + q); break; }
					e = q < n ? (const unsigned char*)memchr(w.data() + q, '\n', n - q) : 0;
				}
			}
			cut.push_back(size);
		}
		const int np = (int)cut.size() - 1;
		if (np < 2) { close(fd); return false; }
		std::mutex mu; std::condition_variable cv; int next = 0, delivered = 0; bool abort = false;
		const int window = 2 * T;
		auto work = [&]() {
			for (;;) {
				int i;
				{ std::unique_lock<std::mutex> l(mu); i = next++; if (i >= np) return; cv.wait(l, [&] { return abort || i < delivered + window; }); if (abort) return; }
				if (ch.is_dead()) { std::lock_guard<std::mutex> l(mu); abort = true; cv.notify_all(); return; }
				std::vector<blk_t> out;
				fq_reader_t rd(fd, cut[(size_t)i], cut[(size_t)i + 1], keep_comment);
				const bool cut_short = drain(rd, per_block, [&](blk_t b) { out.push_back(std::move(b)); });
				{ std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return abort || delivered == i; }); if (abort) return; }
				if (cut_short && i + 1 < np) {
					{ std::lock_guard<std::mutex> l(mu); abort = true; cv.notify_all(); }
					if (getenv("SSG_DEBUG")) fprintf(stderr, "[fastq] %s: not one record per four lines near byte %zu; one thread from there\n", path.c_str(), cut[(size_t)i + 1]);
					out.clear();
					fq_reader_t rest(fd, cut[(size_t)i], size, keep_comment);
					(void)drain(rest, per_block, [&](blk_t b) { ch.push(std::move(b)); });
					return;
				}
				for (blk_t &b : out) ch.push(std::move(b));
				{ std::lock_guard<std::mutex> l(mu); delivered = i + 1; cv.notify_all(); }
			}
		};
		std::vector<std::thread> w;
		for (int t = 0; t < std::min(T, np); ++t) w.emplace_back(work);
		for (std::thread &x : w) x.join();
		close(fd);
		return true;
	}
	/* A compressed file whose decoder outruns one parsing thread (fast_inflate_mt.h): the decoded chunks are gathered into slabs, a slab
	 * is cut at lines that look like record starts and its pieces go through the kseq grammar on several threads, delivered in order.
	 * Same induction as parse_plain(): the first piece starts at a record start (the stream's first byte, or the cut the previous slab
	 * ended on: the bytes after a slab's last cut are not parsed with it but carried into the next, so every parsed piece is closed by a
	 * cut); a piece that runs out of bytes inside a quality string shows that its closing cut was no record start -- everything
	 * from that piece's first byte on, the rest of the stream included, then goes through one thread. */
	void parse_stream(fq_stream_t &src, bool keep_comment, int per_block, int T)
	{
		typedef fq_stream_t::chunk_t chunk_t;
		size_t SLAB = (size_t)64 << 20, PIECE = (size_t)4 << 20;
		{ const char *e = getenv("SSG_FASTQ_SLAB"); if (e && atol(e) > 0) SLAB = (size_t)atol(e); }
		{ const char *e = getenv("SSG_FASTQ_PIECE"); if (e && atol(e) > 0) PIECE = (size_t)atol(e); }
		std::vector<unsigned char> slab[2], carry; int cur = 0;
		std::vector<size_t> cut[2];
		std::vector<std::thread> workers;
		std::mutex mu; std::condition_variable cv; int next = 0, delivered = 0, np_now = 0; bool abort = false; long fb_piece = -1;
		bool eof = false;
		auto serial_from = [&](std::vector<unsigned char> &P) {   /* P, then whatever the decoder still delivers, by one thread */
			fq_reader_t rest(P.data(), P.size(), &src, keep_comment);
			(void)drain(rest, per_block, [this](blk_t b) { ch.push(std::move(b)); });
		};
		auto join_prev = [&]() { for (std::thread &x : workers) x.join(); workers.clear(); };
		while (!eof) {
			std::vector<unsigned char> &S = slab[cur]; std::vector<size_t> &C = cut[cur];
			S.clear(); S.insert(S.end(), carry.begin(), carry.end()); carry.clear();
			const size_t target = std::max(SLAB, S.size() + 1);             /* a slab that ended without a cut comes back as carry: it must grow */
			while (S.size() < target) {
				std::unique_ptr<chunk_t> c;
				if (!src.full.pop(c)) { eof = true; break; }
				S.insert(S.end(), c->begin(), c->end());
				{ std::lock_guard<std::mutex> l(src.empty.mu); if (src.empty.q.size() < 8) src.empty.q.push_back(std::move(c)); }
			}
			join_prev();                                                   /* the other slab's pieces are all delivered (or given up) now */
			if (fb_piece >= 0 || ch.is_dead()) {
				if (fb_piece >= 0 && !ch.is_dead()) {
					const std::vector<unsigned char> &Sp = slab[cur ^ 1]; const std::vector<size_t> &Cp = cut[cur ^ 1];
					if (getenv("SSG_DEBUG")) fprintf(stderr, "[fastq] not one record per four lines in the decoded stream; one thread from there\n");
					std::vector<unsigned char> P(Sp.begin() + (long)Cp[(size_t)fb_piece], Sp.begin() + (long)Cp.back());   /* what the other slab had not delivered; its tail already heads S */
					P.insert(P.end(), S.begin(), S.end());
					serial_from(P);
				}
				return;
			}
			/* cuts: 0, then the first line that looks like a record start at or after every PIECE bytes */
			const unsigned char *p = S.data(); const size_t n = S.size();
			C.assign(1, 0);
			for (size_t pos = PIECE; pos < n; pos += PIECE) {
				const unsigned char *e = (const unsigned char*)memchr(p + pos, '\n', std::min(n - pos, (size_t)1 << 18));
				while (e) {
					const size_t q = (size_t)(e - p) + 1;
					if (q >= n) break;
					if (looks_like_record(p, n, q)) { if (q > C.back()) C.push_back(q); break; }
					e = q - pos < ((size_t)1 << 18) ? (const unsigned char*)memchr(p + q, '\n', std::min(n - q, ((size_t)1 << 18) - (q - pos))) : 0;
				}
			}
			if (eof) C.push_back(n);
			else {
				if (C.size() < 2) {                                         /* nothing that looks like a record start: not a four-line file; one thread from here */
					if (S.size() < 4 * SLAB) { carry.swap(S); continue; }
					serial_from(S); return;
				}
				carry.assign(S.begin() + (long)C.back(), S.end());          /* parsed with the next slab */
			}
			np_now = (int)C.size() - 1; next = 0; delivered = 0; abort = false;
			const bool last_slab = eof;
			const unsigned char *base = S.data(); const std::vector<size_t> *Cc = &C;
			auto work = [&, base, Cc, last_slab]() {
				for (;;) {
					int i;
					{ std::lock_guard<std::mutex> l(mu); i = next++; if (i >= np_now || abort) return; }
					std::vector<blk_t> out;
					fq_reader_t rd(base + (*Cc)[(size_t)i], (*Cc)[(size_t)i + 1] - (*Cc)[(size_t)i], 0, keep_comment);
					const bool cut_short = drain(rd, per_block, [&](blk_t b) { out.push_back(std::move(b)); });
					{ std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return abort || delivered == i; }); if (abort) return; }
					if (cut_short && !(last_slab && i + 1 == np_now)) { std::lock_guard<std::mutex> l(mu); abort = true; fb_piece = i; cv.notify_all(); return; }
					for (blk_t &b : out) ch.push(std::move(b));
					{ std::lock_guard<std::mutex> l(mu); delivered = i + 1; if (ch.is_dead()) abort = true; cv.notify_all(); }
				}
			};
			for (int t = 0; t < std::min(T, np_now); ++t) workers.emplace_back(work);
			cur ^= 1;
		}
		join_prev();
		if (fb_piece >= 0 && !ch.is_dead()) {                              /* in the last slab: the rest of it by one thread */
			const std::vector<unsigned char> &Sp = slab[cur ^ 1]; const std::vector<size_t> &Cp = cut[cur ^ 1];
			std::vector<unsigned char> P(Sp.begin() + (long)Cp[(size_t)fb_piece], Sp.end());
			serial_from(P);
		}
		if (src.had_io_err() && !ch.is_dead()) { blk_t b = fresh(1); b->err = -2; ch.push(std::move(b)); }   /* the decoder gave up: the file is malformed, not merely at its end */
	}
	/* threads_hint: parse threads when the caller knows better than the default (several GPUs to feed); SSG_FASTQ_THREADS wins */
	fq_feed_t(gzFile fp, bool keep_comment, int per_block, const char *path = 0, int threads_hint = 0) : ch(4), pool(new fq_block_pool_t())
	{
		const std::string pth(path ? path : "");
		th = std::thread([this, fp, keep_comment, per_block, pth, threads_hint]() {
			if (pth.empty() || !parse_plain(pth, keep_comment, per_block, threads_hint)) {
				fq_stream_t src(fp, pth.empty() ? 0 : pth.c_str());
				int T = threads_hint > 0 ? threads_hint : std::thread::hardware_concurrency() >= 32 ? 4 : 1;
				{ const char *e = getenv("SSG_FASTQ_THREADS"); if (e) T = atoi(e); }
				if (src.threaded && T > 1) parse_stream(src, keep_comment, per_block, T);
				else { fq_reader_t rd((const unsigned char*)0, 0, &src, keep_comment); (void)drain(rd, per_block, [this](blk_t b) { ch.push(std::move(b)); }); }
			}
			ch.close();
		});
	}
	fq_feed_t(fq_stream_t::provider_t prov, bool keep_comment, int per_block, int threads_hint = 0) : ch(4), pool(new fq_block_pool_t())
	{
		th = std::thread([this, prov, keep_comment, per_block, threads_hint]() {
			{	fq_stream_t src(prov);
				int T = threads_hint > 0 ? threads_hint : std::thread::hardware_concurrency() >= 32 ? 4 : 2;
				{ const char *e = getenv("SSG_FASTQ_THREADS"); if (e) T = atoi(e); }
				if (T > 1) parse_stream(src, keep_comment, per_block, T);
				else { fq_reader_t rd((const unsigned char*)0, 0, &src, keep_comment); (void)drain(rd, per_block, [this](blk_t b) { ch.push(std::move(b)); }); }
			}
			ch.close();
		});
	}
	/* one serial reader made by the caller (bwa_main.cpp: the parser taking over from the device-text path in the middle of an input) */
	fq_feed_t(std::function<fq_reader_t*()> mk, int per_block) : ch(4), pool(new fq_block_pool_t())
	{
		th = std::thread([this, mk, per_block]() {
			{ std::unique_ptr<fq_reader_t> rd(mk()); if (rd) (void)drain(*rd, per_block, [this](blk_t b) { ch.push(std::move(b)); }); }
			ch.close();
		});
	}
	~fq_feed_t() { ch.abandon(); if (th.joinable()) th.join(); }
};

/* sequential view over a feed: one record at a time */
struct fq_cursor_t {
	fq_feed_t &feed; std::shared_ptr<fq_block_t> cur; int i;
	explicit fq_cursor_t(fq_feed_t &f) : feed(f), i(0) {}
	/* 0: record (*blk, *idx) available; -1 EOF; -2 malformed */
	int next(const fq_block_t **blk, int *idx)
	{
		while (!cur || i >= cur->n) {
			if (cur && cur->err) return cur->err;
			std::shared_ptr<fq_block_t> nb;
			if (!feed.ch.pop(nb)) return -1;
			cur = std::move(nb); i = 0;
			if (cur->n == 0 && cur->err) return cur->err;
		}
		*blk = cur.get(); *idx = i++;
		return 0;
	}
};
#endif
