/*
 * ranksplit.h -- rank mode (ranks.h): who reads which bytes of the input.
 *
 * Every rank's `bwa mem` could parse the whole FASTQ and keep the batches of its rank (it does, for compressed input and pipes) -- N parsers of
 * everything, each at the rate of one, is a ceiling no number of GPUs lifts.  For plain regular files rank 0 scans instead: records found by
 * their newlines (four lines each, checked line by line; anything else stops the run with a message that names the switch), upstream's
 * batches formed from the sequence lengths exactly as bseq_read forms them (bwa.c: a batch ends once its bases reach the chunk size, on an
 * even read count; interleaved or two files in step), one line of five numbers per batch appended to SSG_RDV/batches: pairs, and the byte range
 * of the batch in each file.  Every rank -- rank 0 too -- then parses only the ranges of its batches, handed to the parser as a stream
 * (fastq.h provider mode).  A batch cut out of the file is a batch to the rank's own bseq_read logic as well: each starts the base count
 * afresh, so the boundaries it finds in its stream are the scanner's.
 */
#ifndef SSG_RANKSPLIT_H
#define SSG_RANKSPLIT_H
#include "ranks.h"
#include "fastq.h"
#include <mutex>

struct rs_entry_t { uint64_t pairs, a0, a1, b0, b1; };

static inline bool rs_plain_regular(const char *path)
{
	struct stat sb; unsigned char m[2];
	if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 2) return false;
	const int fd = open(path, O_RDONLY); if (fd < 0) return false;
	const bool plain = pread(fd, m, 2, 0) == 2 && !(m[0] == 0x1f && m[1] == 0x8b);
	close(fd); return plain;
}

/* one input file, read front to back in large pieces: the next four-line record's byte range and sequence length */
struct rs_scan_t {
	int fd; size_t size, base, fill; std::vector<unsigned char> buf; size_t at;   /* buf[0 .. fill) = file bytes [base, base + fill); at = scan position in the file */
	std::string why;
	explicit rs_scan_t(const char *path) : fd(open(path, O_RDONLY)), size(0), base(0), fill(0), buf((size_t)16 << 20), at(0) { struct stat sb; if (fd >= 0 && fstat(fd, &sb) == 0) size = (size_t)sb.st_size; }
	~rs_scan_t() { if (fd >= 0) close(fd); }
	bool more(size_t need_from)
	{	/* keep [need_from, ...) and read on */
		const size_t keep = base + fill - need_from;
		if (keep == buf.size()) buf.resize(buf.size() * 2);
		memmove(buf.data(), buf.data() + (need_from - base), keep); base = need_from; fill = keep;
		while (base + fill < size) { const ssize_t r = pread(fd, buf.data() + fill, buf.size() - fill, (off_t)(base + fill)); if (r < 0 && errno == EINTR) continue; if (r <= 0) break; fill += (size_t)r; return true; }
		return false;
	}
	/* end of the line that starts at file offset p (offset of its '\n', or of the end of the file); false: cannot tell yet (never: reads on) */
	size_t line_end(size_t p, size_t rec0)
	{
		for (;;) {
			if (p < base + fill) { const unsigned char *e = (const unsigned char*)memchr(buf.data() + (p - base), '\n', base + fill - p); if (e) return base + (size_t)(e - buf.data()); }
			if (base + fill >= size) return size;
			if (!more(rec0)) return size;
		}
	}
	/* 1: record [*r0, *r1) with *len bases; 0: end of the file; -1: not four lines per record (why) */
	int next(size_t *r0, size_t *r1, size_t *len)
	{
		for (;;) {   /* blank space after the last record is what kseq skips too */
			if (at >= size) return 0;
			if (at >= base + fill && !more(at)) return 0;
			const unsigned char c = buf[at - base];
			if (c == '\n' || c == '\r' || c == ' ' || c == '\t') { ++at; continue; }
			break;
		}
		const size_t rec0 = at;
		if (buf[at - base] != '@') { why = "a record that does not start with '@'"; return -1; }
		const size_t e1 = line_end(at, rec0); if (e1 >= size) { why = "a header line without a sequence"; return -1; }
		const size_t s0 = e1 + 1, e2 = line_end(s0, rec0); if (e2 >= size) { why = "a sequence without a '+' line"; return -1; }
		const size_t p0 = e2 + 1;
		if (p0 >= base + fill && !more(rec0)) { why = "a sequence without a '+' line"; return -1; }
		if (buf[p0 - base] != '+') { why = "a sequence of several lines (or no quality line)"; return -1; }
		const size_t e3 = line_end(p0, rec0); if (e3 >= size) { why = "a '+' line without qualities"; return -1; }
		const size_t q0 = e3 + 1, e4 = line_end(q0, rec0);
		if (e4 - q0 != e2 - s0) { why = "a quality string that is not as long as its sequence"; return -1; }
		if (e2 > s0 && buf[e2 - 1 - base] == '\r') { why = "lines that end in CR LF"; return -1; }
		if (e2 == s0) { why = "an empty sequence"; return -1; }
		*r0 = rec0; *r1 = e4 < size ? e4 + 1 : size; *len = e2 - s0; at = *r1;
		return 1;
	}
};

/* rank 0: scan and publish.  Returns false (message printed, SSG_RDV/batches.fail written) when the input is not four lines per record. */
static inline bool rs_scan_and_publish(const std::string &rdv, const char *f1, const char *f2, int64_t chunk)
{
	rs_scan_t A(f1); std::unique_ptr<rs_scan_t> B(f2 ? new rs_scan_t(f2) : 0);
	const std::string path = rdv + "/batches";
	const int out = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_APPEND, 0644);
	auto fail = [&](const std::string &msg) { fprintf(stderr, "[bwa] rank mode: %s; SSG_RANKS_SPLIT=0 makes every rank parse the whole input instead\n", msg.c_str()); (void)rk_file_put(rdv + "/batches.fail", msg.data(), msg.size()); if (out >= 0) close(out); return false; };
	if (A.fd < 0 || (B && B->fd < 0) || out < 0) return fail("cannot open the input or the rendezvous directory");
	uint64_t n_batches = 0; bool eof = false;
	while (!eof) {
		rs_entry_t e; e.pairs = 0; e.a0 = A.at; e.b0 = B ? B->at : 0; e.a1 = e.a0; e.b1 = e.b0;
		int64_t bases = 0; bool first = true;
		for (;;) {
			size_t r0, r1, l0, l1;
			int rc = A.next(&r0, &r1, &l0);
			if (rc < 0) return fail(std::string(f1) + " has " + A.why);
			if (rc == 0) { eof = true; break; }
			if (first) e.a0 = r0;
			size_t m0 = 0, m1 = 0;
			rs_scan_t &S2 = B ? *B : A;
			const size_t a1_before = r1;
			rc = S2.next(&m0, &m1, &l1);
			if (rc < 0) return fail(std::string(B ? f2 : f1) + " has " + S2.why);
			if (rc == 0) { fprintf(stderr, B ? "[W::bseq_read] the 2nd file has fewer sequences.\n" : "[W::main_mem] odd number of reads in the PE mode; last read dropped\n"); eof = true; break; }   /* upstream: the read without a mate is dropped */
			if (B) { if (first) e.b0 = m0; e.a1 = a1_before; e.b1 = m1; } else e.a1 = m1;
			first = false;
			++e.pairs; bases += (int64_t)l0 + (int64_t)l1;
			if (bases >= chunk) break;
		}
		if (e.pairs) { if (write(out, &e, sizeof(e)) != (ssize_t)sizeof(e)) return fail("cannot write into the rendezvous directory"); ++n_batches; }
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
	{
		std::lock_guard<std::mutex> l(mu);
		const double t0 = rk_now(); struct stat sb;
		for (;;) {
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
			if (rk_now() - t0 > rk_timeout() || rk_someone_failed()) return -1;
			usleep(5000);
		}
	}
};

/* the bytes of this rank's batches in one of the files, as a stream for the parser */
static inline fq_stream_t::provider_t rs_provider(std::shared_ptr<rs_table_t> tab, const char *path, bool second_file, int rank, int world, std::atomic<int> *failed)
{
	struct st_t { int fd; uint64_t k, off, end; bool open_range; };
	std::shared_ptr<st_t> st(new st_t()); st->fd = open(path, O_RDONLY); st->k = 0; st->off = st->end = 0; st->open_range = false;
	return [tab, st, second_file, rank, world, failed](fq_stream_t::chunk_t &c) -> bool {
		if (st->fd < 0) { failed->store(1); return false; }
		while (!st->open_range || st->off >= st->end) {
			rs_entry_t e; uint64_t id0;
			const int rc = tab->get((uint64_t)rank + st->k * (uint64_t)world, &e, &id0);
			if (rc < 0) { failed->store(1); return false; }
			if (rc == 0) return false;
			++st->k; st->off = second_file ? e.b0 : e.a0; st->end = second_file ? e.b1 : e.a1; st->open_range = true;
		}
		const size_t want = (size_t)std::min<uint64_t>(st->end - st->off, (uint64_t)4 << 20);
		c.resize(want);
		for (size_t got = 0; got < want; ) { const ssize_t r = pread(st->fd, c.data() + got, want - got, (off_t)(st->off + got)); if (r < 0 && errno == EINTR) continue; if (r <= 0) { failed->store(1); return false; } got += (size_t)r; }
		st->off += want;
		return true;
	};
}
#endif
