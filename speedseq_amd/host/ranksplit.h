/*
 * ranksplit.h -- rank mode (ranks.h): who reads which bytes of the input.
 *
 * Every rank's `bwa mem` could parse the whole FASTQ and keep the batches of its rank (it does, for compressed input and pipes) -- N parsers of
 * everything, each at the rate of one, is a ceiling no number of GPUs lifts.  For plain regular files rank 0 scans instead: records found by
 * their newlines (four lines each, checked line by line; anything else stops the run with a message that names the switch), upstream's
 * batches formed from the sequence lengths exactly as bseq_read forms them (bwa.c: a batch ends once its bases reach the chunk size, on an
 * even read count; interleaved or two files in step), one line of five numbers per batch appended to SSG_RDV/batches: pairs, and the byte range
 * of the batch in each file.  Every rank -- rank 0 too -- then parses only the ranges of its batches, handed to the parser as a stream
 * (fastq.h provider mode).  Compressed input is inflated by rank 0 alone (one gzip stream has one reader however many want its bytes): the
 * scanner runs behind fastq.h's decoder threads and hands every batch on as a file of its own in SSG_RDV.  A batch cut out of the file is a batch to the rank's own bseq_read logic as well: each starts the base count
 * afresh, so the boundaries it finds in its stream are the scanner's.
 */
#ifndef SSG_RANKSPLIT_H
#define SSG_RANKSPLIT_H
#include "ranks.h"
#include "fastq.h"
#include <mutex>
#include <functional>
#include <thread>

struct rs_entry_t { uint64_t pairs, a0, a1, b0, b1; };

static inline bool rs_plain_regular(const char *path)
{
	struct stat sb; unsigned char m[2];
	if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 2) return false;
	const int fd = open(path, O_RDONLY); if (fd < 0) return false;
	const bool plain = pread(fd, m, 2, 0) == 2 && !(m[0] == 0x1f && m[1] == 0x8b);
	close(fd); return plain;
}

/* one input, read front to back: the next four-line record's byte range (in bytes of the decoded input) and sequence length.  The bytes come from
 * a sequential reader: pread over a plain file, or the decoder threads of fastq.h for compressed input */
struct rs_scan_t {
	std::function<long(unsigned char*, size_t)> rd; bool eof_seen;
	size_t base, fill; std::vector<unsigned char> buf; size_t at;   /* buf[0 .. fill) = input bytes [base, base + fill); at = scan position */
	std::string why;
	std::function<void()> before_move;   /* called before bytes already handed out through ptr() move or go away (a caller that still points at them copies them now) */
	explicit rs_scan_t(std::function<long(unsigned char*, size_t)> r, size_t from = 0) : rd(r), eof_seen(false), base(from), fill(0), buf((size_t)16 << 20), at(from) {}   /* from: the reader starts at this offset of the input */
	size_t lim() const { return eof_seen ? base + fill : (size_t)-1; }       /* the end of the input, once it has been seen */
	const unsigned char *ptr(size_t off) const { return buf.data() + (off - base); }   /* valid for offsets >= the last record's start, until the next call */
	bool more(size_t need_from)
	{	/* keep [need_from, ...) and read on */
		if (before_move) before_move();
		const size_t keep = base + fill - need_from;
		if (keep == buf.size()) buf.resize(buf.size() * 2);
		memmove(buf.data(), buf.data() + (need_from - base), keep); base = need_from; fill = keep;
		if (eof_seen) return false;
		const long r = rd(buf.data() + fill, buf.size() - fill);
		if (r <= 0) { eof_seen = true; return false; }
		fill += (size_t)r; return true;
	}
	/* end of the line that starts at offset p: the offset of its '\n', or lim() when the input ends first */
	size_t line_end(size_t p, size_t rec0)
	{
		for (;;) {
			if (p < base + fill) { const unsigned char *e = (const unsigned char*)memchr(buf.data() + (p - base), '\n', base + fill - p); if (e) return base + (size_t)(e - buf.data()); }
			if (!more(rec0)) return lim();
		}
	}
	/* 1: record [*r0, *r1) with *len bases; 0: end of the input; -1: not four lines per record (why) */
	int next(size_t *r0, size_t *r1, size_t *len)
	{
		for (;;) {   /* blank space after the last record is what kseq skips too */
			if (at >= base + fill && !more(at)) return 0;
			const unsigned char c = buf[at - base];
			if (c == '\n' || c == '\r' || c == ' ' || c == '\t') { ++at; continue; }
			break;
		}
		const size_t rec0 = at;
		if (buf[at - base] != '@') { why = "a record that does not start with '@'"; return -1; }
		const size_t e1 = line_end(at, rec0); if (e1 >= lim()) { why = "a header line without a sequence"; return -1; }
		const size_t s0 = e1 + 1, e2 = line_end(s0, rec0); if (e2 >= lim()) { why = "a sequence without a '+' line"; return -1; }
		const size_t p0 = e2 + 1;
		if (p0 >= base + fill && !more(rec0)) { why = "a sequence without a '+' line"; return -1; }
		if (buf[p0 - base] != '+') { why = "a sequence of several lines (or no quality line)"; return -1; }
		const size_t e3 = line_end(p0, rec0); if (e3 >= lim()) { why = "a '+' line without qualities"; return -1; }
		const size_t q0 = e3 + 1, e4 = line_end(q0, rec0);
		if (e4 - q0 != e2 - s0) { why = "a quality string that is not as long as its sequence"; return -1; }
		if (e2 > s0 && buf[e2 - 1 - base] == '\r') { why = "lines that end in CR LF"; return -1; }
		if (e2 == s0) { why = "an empty sequence"; return -1; }
		*r0 = rec0; *r1 = e4 < lim() ? e4 + 1 : lim(); *len = e2 - s0; at = *r1;
		return 1;
	}
};
static inline std::function<long(unsigned char*, size_t)> rs_file_reader(const char *path, size_t from = 0)
{
	struct st_t { int fd; size_t off; ~st_t() { if (fd >= 0) close(fd); } };
	std::shared_ptr<st_t> st(new st_t()); st->fd = open(path, O_RDONLY); st->off = from;
	return [st](unsigned char *d, size_t cap) -> long {
		for (;;) { const ssize_t r = st->fd < 0 ? -1 : pread(st->fd, d, cap, (off_t)st->off); if (r < 0 && errno == EINTR) continue; if (r > 0) st->off += (size_t)r; return (long)r; }
	};
}
/* compressed input (or a pipe): fastq.h's stream with its decoder threads; *bad is set when the decoder met a damaged stream */
static inline std::function<long(unsigned char*, size_t)> rs_stream_reader(const char *path, std::atomic<int> *bad)
{
	struct st_t { gzFile fp; std::unique_ptr<fq_stream_t> ks; ~st_t() { ks.reset(); if (fp) gzclose(fp); } };
	std::shared_ptr<st_t> st(new st_t()); st->fp = gzopen(path, "r");
	if (st->fp) st->ks.reset(new fq_stream_t(st->fp, path));
	return [st, bad](unsigned char *d, size_t cap) -> long {
		if (!st->ks) return -1;
		fq_stream_t &ks = *st->ks;
		if (ks.begin >= ks.end && !ks.fill()) { if (ks.had_io_err()) bad->store(1); return 0; }
		const size_t k = std::min(cap, (size_t)(ks.end - ks.begin));
		memcpy(d, ks.buf.data() + ks.begin, k); ks.begin += (int)k;
		return (long)k;
	};
}

/* The same scan of a plain regular file by several threads.  One thread finding four newlines per record gets 2.8 GB/s out of the page cache (4.5 M pairs/s of
 * 2x150: the one stage of rank mode that did not divide by the ranks); here the file is mapped and cut into slices, a thread each.  A slice's thread guesses where its
 * first record starts (a line that begins two plain records in a row) and scans plain records -- '@' line, sequence, '+' line, qualities as long as the sequence, each
 * ended by '\n', no CR, no blank space between records -- until one starts beyond its slice or is not plain.  The consumer walks the slices in order and takes a slice's
 * records only if its guess is exactly where the slice before ended (else that slice is scanned again from there); at the first record that is not plain it hands the
 * rest of the file to rs_scan_t, which accepts or refuses it with its own words.  What next() returns is therefore what rs_scan_t returns, record for record. */
struct rs_pscan_t {
	size_t size; int fd; std::string path;
	size_t at; std::string why;
	int n_thr; size_t slice;
	static constexpr size_t MARGIN = (size_t)1 << 20;   /* bytes read beyond a slice: the record that starts in it ends there (a longer one goes to the one-thread scanner) */
	struct sl_t { size_t guess, end; bool plain_end, checked; std::vector<uint32_t> rec; std::vector<unsigned char> buf; size_t base, fill; };   /* rec: record length, sequence length per record from `guess`; end: where the next record starts (or what is not a plain record); buf: bytes [base, base + fill) of the file */
	std::vector<sl_t> win; size_t w0;          /* slices of the current window, which starts at file offset w0 */
	size_t cur_sl, cur_rec;
	std::unique_ptr<rs_scan_t> slow;
	std::function<void()> before_move;   /* as rs_scan_t's: called before a window's buffers are loaded anew or handed back */
	void to_slow(size_t from) { if (before_move) before_move(); slow.reset(new rs_scan_t(rs_file_reader(path.c_str(), from), from)); slow->before_move = before_move; win.clear(); cur_sl = 0; }
	/* the bytes of the record next() has just returned (valid until the next call, or until before_move is called) */
	const unsigned char *ptr(size_t off) const { if (slow) return slow->ptr(off); const sl_t &S = win[cur_sl]; return S.buf.data() + (off - S.base); }
	explicit rs_pscan_t(const char *p) : size(0), fd(-1), path(p), at(0), n_thr(1), slice((size_t)32 << 20), w0(0), cur_sl(0), cur_rec(0)
	{
		struct stat sb;
		fd = open(p, O_RDONLY);
		if (fd >= 0 && fstat(fd, &sb) == 0 && sb.st_size > 0) size = (size_t)sb.st_size;
		const unsigned hw = std::thread::hardware_concurrency();
		n_thr = (int)std::max(1u, std::min(8u, hw ? hw / 2 : 1u));
		{ const char *e = getenv("SSG_RANKS_SCAN_THREADS"); if (e && atoi(e) > 0) n_thr = std::min(64, atoi(e)); }
		{ const char *e = getenv("SSG_RANKS_SCAN_SLICE"); if (e && atol(e) > 0) slice = (size_t)atol(e); }   /* tests: many slices in a small file */
		if (!size) slow.reset(new rs_scan_t(rs_file_reader(p)));   /* cannot be opened (or empty): the one-thread scanner says what there is to say (a caller sets before_move after construction: nothing has been handed out yet) */
	}
	~rs_pscan_t() { if (fd >= 0) close(fd); }
	/* bytes [from, from + want) of the file (less at its end) into the slice's buffer */
	bool load(sl_t &S, size_t from, size_t want) const
	{
		if (S.buf.size() < want) S.buf.resize(want);
		S.base = from; S.fill = 0;
		while (S.fill < want && from + S.fill < size) {
			const ssize_t r = pread(fd, S.buf.data() + S.fill, std::min(want - S.fill, size - from - S.fill), (off_t)(from + S.fill));
			if (r < 0 && errno == EINTR) continue;
			if (r <= 0) return false;
			S.fill += (size_t)r;
		}
		return true;
	}
	/* a plain record at file offset p (inside the buffer): its length and sequence length; 0 when what starts at p is not one, or does not end inside the buffer */
	static size_t plain(const sl_t &S, size_t p, size_t size, uint32_t *seqlen)
	{
		const unsigned char *d = S.buf.data() - S.base; const size_t end = S.base + S.fill;   /* d[offset] for offsets in [base, end) */
		if (p >= end || d[p] != '@') return 0;
		const unsigned char *e1 = (const unsigned char*)memchr(d + p, '\n', end - p); if (!e1) return 0;
		const size_t s0 = (size_t)(e1 - d) + 1;
		const unsigned char *e2 = s0 < end ? (const unsigned char*)memchr(d + s0, '\n', end - s0) : 0; if (!e2) return 0;
		const size_t L = (size_t)(e2 - d) - s0, p0 = (size_t)(e2 - d) + 1;
		if (!L || d[s0 + L - 1] == '\r' || p0 >= end || d[p0] != '+') return 0;
		size_t e3;
		if (p0 + 1 < end && d[p0 + 1] == '\n') e3 = p0 + 1;
		else { const unsigned char *x = (const unsigned char*)memchr(d + p0, '\n', end - p0); if (!x) return 0; e3 = (size_t)(x - d); }
		const size_t q0 = e3 + 1;
		if (q0 + L >= end) return 0;   /* (at the end of the file: the last record without a final newline is the one-thread scanner's) */
		const unsigned char *e4 = (const unsigned char*)memchr(d + q0, '\n', L + 1);
		if (!e4 || (size_t)(e4 - d) - q0 != L || L > 0xfffffff0u) return 0;
		*seqlen = (uint32_t)L;
		const size_t len = (size_t)(e4 - d) + 1 - p;
		(void)size;
		return len > 0xfffffff0u ? 0 : len;
	}
	void scan_slice(sl_t &S, size_t from, size_t lim) const
	{	/* plain records from `from` until one starts at or beyond lim; the buffer holds [from-ish, lim + MARGIN) */
		S.rec.clear(); S.guess = from; S.plain_end = true; S.checked = false;
		size_t p = from; uint32_t L = 0;
		while (p < lim) {
			const size_t n = plain(S, p, size, &L);
			if (!n) { S.plain_end = false; break; }
			S.rec.push_back((uint32_t)n); S.rec.push_back(L); p += n;
		}
		S.end = p;
	}
	size_t guess_start(const sl_t &S, size_t b, size_t lim) const
	{	/* the first line start in (b - 1, lim) that begins two plain records in a row (lim when there is none among the first few lines); the buffer starts at b - 1 */
		const unsigned char *d = S.buf.data() - S.base; const size_t end = S.base + S.fill;
		const unsigned char *nl = (const unsigned char*)memchr(d + b - 1, '\n', end - (b - 1));   /* a record can start at b only behind a newline */
		size_t c = nl ? (size_t)(nl - d) + 1 : lim;
		for (int k = 0; k < 8 && c < lim; ++k) {
			uint32_t L; const size_t n = plain(S, c, size, &L);
			if (n && (c + n >= size || plain(S, c + n, size, &L))) return c;
			const unsigned char *x = (const unsigned char*)memchr(d + c, '\n', end - c);
			c = x ? (size_t)(x - d) + 1 : lim;
		}
		return lim;
	}
	bool next_window()
	{	/* slices from `at` (a record starts there: slice 0 needs no guess), a thread each: read, guess, scan */
		if (before_move) before_move();
		w0 = at; int k = 0;
		while (k < n_thr && w0 + (size_t)k * slice < size) ++k;
		win.resize((size_t)k);
		std::atomic<int> io_bad(0);
		std::vector<std::thread> th;
		for (int i = 0; i < k; ++i) th.emplace_back([this, i, &io_bad]() {
			sl_t &S = win[(size_t)i];
			const size_t b = w0 + (size_t)i * slice, e = std::min(size, b + slice), from = i == 0 ? b : b - 1;
			if (!load(S, from, e - from + MARGIN)) { io_bad = 1; return; }
			scan_slice(S, i == 0 ? b : guess_start(S, b, e), e);
		});
		for (auto &t : th) t.join();
		cur_sl = 0; cur_rec = 0;
		return !io_bad.load();
	}
	/* as rs_scan_t::next */
	int next(size_t *r0, size_t *r1, size_t *len)
	{
		for (;;) {
			if (slow) { if (!slow->before_move && before_move) slow->before_move = before_move; const int rc = slow->next(r0, r1, len); at = slow->at; if (rc < 0) why = slow->why; return rc; }
			if (cur_sl >= win.size()) {
				if (at >= size || !next_window()) { to_slow(at); continue; }   /* the end of the input (or a read error): in the one-thread scanner's words */
			}
			sl_t &S = win[cur_sl];
			if (!S.checked) {
				const size_t lim = std::min(size, w0 + slice * (cur_sl + 1));
				if (at >= lim) { ++cur_sl; cur_rec = 0; continue; }   /* the records before ran over this whole slice */
				if (S.guess != at) {                                 /* the guess was not where the records before end: this slice again, from there */
					if (at < S.base && !load(S, at, lim - at + MARGIN)) { to_slow(at); continue; }
					scan_slice(S, at, lim);
				}
				S.checked = true; cur_rec = 0;
			}
			if (cur_rec < S.rec.size()) {
				*r0 = at; *r1 = at + S.rec[cur_rec]; *len = S.rec[cur_rec + 1]; cur_rec += 2; at = *r1;
				return 1;
			}
			if (!S.plain_end) { to_slow(at); continue; }   /* something that is not a plain record: from here on one thread, and its words */
			++cur_sl; cur_rec = 0;
		}
	}
};

/* rank 0: scan and publish.  Plain regular files: the batches' byte ranges in the files.  Anything else (`served`: compressed input, which only one
 * process should inflate): the batches' bytes themselves, a file per batch and input in SSG_RDV (fq.<batch>.<1|2>, removed by the rank that reads
 * it; rank 0 stays at most three rounds of batches ahead).  Returns false (message printed, SSG_RDV/batches.fail written) when the input is not
 * four lines per record or cannot be read. */
static inline bool rs_scan_and_publish(const std::string &rdv, const char *f1, const char *f2, int64_t chunk, bool served, int world)
{
	std::atomic<int> bad(0);
	/* plain files: the several-thread scanner (SSG_RANKS_SCAN_THREADS=1 with SSG_RANKS_SCAN_MT=0: the one-thread one); served input: the one-thread scanner behind the decoder */
	struct any_t {
		std::unique_ptr<rs_scan_t> s; std::unique_ptr<rs_pscan_t> p;
		int next(size_t *r0, size_t *r1, size_t *len) { return p ? p->next(r0, r1, len) : s->next(r0, r1, len); }
		size_t at() const { return p ? p->at : s->at; }
		const std::string &why() const { return p ? p->why : s->why; }
		const unsigned char *ptr(size_t off) const { return s->ptr(off); }   /* served input only */
	};
	const bool mt = !served && !(getenv("SSG_RANKS_SCAN_MT") && !strcmp(getenv("SSG_RANKS_SCAN_MT"), "0"));
	auto mk = [&](const char *f) { std::unique_ptr<any_t> a(new any_t()); if (mt) a->p.reset(new rs_pscan_t(f)); else a->s.reset(new rs_scan_t(served ? rs_stream_reader(f, &bad) : rs_file_reader(f))); return a; };
	std::unique_ptr<any_t> Ap = mk(f1), B(f2 ? mk(f2).release() : 0);
	any_t &A = *Ap;
	const std::string path = rdv + "/batches";
	const int out = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_APPEND, 0644);
	auto fail = [&](const std::string &msg) { fprintf(stderr, "[bwa] rank mode: %s; SSG_RANKS_SPLIT=0 makes every rank parse the whole input instead\n", msg.c_str()); (void)rk_file_put(rdv + "/batches.fail", msg.data(), msg.size()); if (out >= 0) close(out); return false; };
	if (out < 0) return fail("cannot write into the rendezvous directory");
	uint64_t n_batches = 0; bool eof = false;
	std::vector<unsigned char> acc1, acc2;
	while (!eof) {
		rs_entry_t e; e.pairs = 0; e.a0 = A.at(); e.b0 = B ? B->at() : 0; e.a1 = e.a0; e.b1 = e.b0;
		int64_t bases = 0; bool first = true;
		acc1.clear(); acc2.clear();
		for (;;) {
			size_t r0, r1, l0, l1;
			int rc = A.next(&r0, &r1, &l0);
			if (rc < 0) return fail(std::string(f1) + " has " + A.why());
			if (rc == 0) { eof = true; break; }
			const size_t keep1 = acc1.size();
			if (served) acc1.insert(acc1.end(), A.ptr(r0), A.ptr(r0) + (r1 - r0));   /* before the next call moves the buffer */
			size_t m0 = 0, m1 = 0;
			any_t &S2 = B ? *B : A;
			rc = S2.next(&m0, &m1, &l1);
			if (rc < 0) return fail(std::string(B ? f2 : f1) + " has " + S2.why());
			if (rc == 0) { fprintf(stderr, B ? "[W::bseq_read] the 2nd file has fewer sequences.\n" : "[W::main_mem] odd number of reads in the PE mode; last read dropped\n"); acc1.resize(keep1); eof = true; break; }   /* upstream: the read without a mate is dropped */
			if (served) { std::vector<unsigned char> &ac = B ? acc2 : acc1; ac.insert(ac.end(), S2.ptr(m0), S2.ptr(m0) + (m1 - m0)); }
			if (first) { e.a0 = r0; if (B) e.b0 = m0; }
			if (B) { e.a1 = r1; e.b1 = m1; } else e.a1 = m1;
			first = false;
			++e.pairs; bases += (int64_t)l0 + (int64_t)l1;
			if (bases >= chunk) break;
		}
		if (bad.load()) return fail("the compressed input is damaged");
		if (!e.pairs) continue;
		if (served) {
			const double t0 = rk_now(); struct stat sb;   /* not more than three rounds ahead of the slowest reader */
			while (n_batches >= 3 * (uint64_t)world && stat((rk_data_dir() + "/fq." + std::to_string(n_batches - 3 * (uint64_t)world) + ".1").c_str(), &sb) == 0) {
				if (rk_someone_failed() || rk_now() - t0 > rk_timeout()) return fail("the other ranks do not take their batches");
				usleep(2000);
			}
			e.a0 = 0; e.a1 = acc1.size(); e.b0 = 0; e.b1 = acc2.size();
			if (!rk_file_put(rk_data_dir() + "/fq." + std::to_string(n_batches) + ".1", acc1.data(), acc1.size()) || (B && !rk_file_put(rk_data_dir() + "/fq." + std::to_string(n_batches) + ".2", acc2.data(), acc2.size()))) return fail("cannot write into the rendezvous directory");
		}
		if (write(out, &e, sizeof(e)) != (ssize_t)sizeof(e)) return fail("cannot write into the rendezvous directory");
		++n_batches;
	}
	close(out);
	return rk_file_put(rdv + "/batches.done", &n_batches, 8);
}

/* every rank: the published batches as they come */
struct rs_table_t {
	std::string rdv; int fd; std::mutex mu; std::vector<rs_entry_t> ent; std::vector<uint64_t> pairs_before; bool done; uint64_t total;
	explicit rs_table_t(const std::string &d) : rdv(d), fd(-1), pairs_before(1, 0), done(false), total(0) {}
	~rs_table_t() { if (fd >= 0) close(fd); }
	/* entry b: 1 = there, 0 = the input has fewer batches, -1 = the scan failed or nobody scans */
	int get(uint64_t b, rs_entry_t *e, uint64_t *id0)
	{	/* the lock is held for a look at the table and a read of what the file has new, never while waiting: a caller whose entry is there
		 * already is not held up by one that polls for an entry yet to be published (with tiny batches the two providers of a rank can be
		 * several batches apart, and rank 0's scanner waits for the slower one) */
		const double t0 = rk_now(); struct stat sb;
		for (;;) {
			{
				std::lock_guard<std::mutex> l(mu);
				if (fd < 0) fd = open((rdv + "/batches").c_str(), O_RDONLY);
				if (fd >= 0) {
					rs_entry_t x;
					while (pread(fd, &x, sizeof(x), (off_t)(ent.size() * sizeof(x))) == (ssize_t)sizeof(x)) { ent.push_back(x); pairs_before.push_back(pairs_before.back() + x.pairs); }
				}
				if (b < ent.size()) { *e = ent[(size_t)b]; *id0 = pairs_before[(size_t)b]; return 1; }
				if (done) return b < total ? -1 : 0;
				if (stat((rdv + "/batches.fail").c_str(), &sb) == 0) return -1;
				std::vector<uint8_t> d;
				if (stat((rdv + "/batches.done").c_str(), &sb) == 0 && rk_file_get(rdv + "/batches.done", d) && d.size() == 8) { memcpy(&total, d.data(), 8); done = true; continue; }   /* once more through the file: the last entries precede the marker */
			}
			if (rk_now() - t0 > rk_timeout() || rk_someone_failed()) return -1;
			usleep(5000);
		}
	}
};

/* the bytes of this rank's batches in one of the files, as a stream for the parser */
static inline fq_stream_t::provider_t rs_provider(std::shared_ptr<rs_table_t> tab, const char *path, bool second_file, int rank, int world, std::atomic<int> *failed, bool served)
{
	struct st_t { int fd; uint64_t k, off, end; bool open_range; std::string cur; };
	std::shared_ptr<st_t> st(new st_t()); st->fd = served ? -1 : open(path, O_RDONLY); st->k = 0; st->off = st->end = 0; st->open_range = false;
	return [tab, st, second_file, rank, world, failed, served](fq_stream_t::chunk_t &c) -> bool {
		if (!served && st->fd < 0) { failed->store(1); return false; }
		while (!st->open_range || st->off >= st->end) {
			if (served && st->fd >= 0) { close(st->fd); st->fd = -1; unlink(st->cur.c_str()); }   /* the batch's file has been read: rank 0 may write on */
			rs_entry_t e; uint64_t id0;
			const uint64_t b = (uint64_t)rank + st->k * (uint64_t)world;
			const int rc = tab->get(b, &e, &id0);
			if (rc < 0) { failed->store(1); return false; }
			if (rc == 0) return false;
			++st->k; st->off = second_file ? e.b0 : e.a0; st->end = second_file ? e.b1 : e.a1; st->open_range = true;
			if (served) { st->cur = rk_data_dir() + "/fq." + std::to_string(b) + (second_file ? ".2" : ".1"); st->fd = open(st->cur.c_str(), O_RDONLY); if (st->fd < 0) { failed->store(1); return false; } }
		}
		const size_t want = (size_t)std::min<uint64_t>(st->end - st->off, (uint64_t)4 << 20);
		c.resize(want);
		for (size_t got = 0; got < want; ) { const ssize_t r = pread(st->fd, c.data() + got, want - got, (off_t)(st->off + got)); if (r < 0 && errno == EINTR) continue; if (r <= 0) { failed->store(1); return false; } got += (size_t)r; }
		st->off += want;
		return true;
	};
}
#endif
