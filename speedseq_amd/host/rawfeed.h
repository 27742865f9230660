/*
 * rawfeed.h -- `bwa mem`'s input as TEXT for the device (csrc/k_bam.h ssg_k_fq_unpack): the host only finds the records and
 * their sequence lengths by their newlines (ranksplit.h's scanners: plain four-line records, checked line by line), forms upstream's batches from
 * the lengths exactly as bseq_read does, and copies the batches' bytes -- in runs as long as the scanner's buffers allow -- into page-locked
 * blocks that travel to the MI355X as they are; names, codes and qualities are taken from the text there.  kseq's record grammar
 * (/root/reference/src/samtools-1.3.1/htslib-1.3.1/htslib/kseq.h:189-229) accepts more than plain records (wrapped sequences, FASTA, CR LF, blank
 * lines): at the first record that is not plain the scanner stops, and everything from the start of the upstream batch it belongs to goes through
 * the parser of fastq.h as before -- the same reads, the same batches, the same output, only slower.
 */
#ifndef SSG_RAWFEED_H
#define SSG_RAWFEED_H
#include "ranksplit.h"
#include "../../include/ssgpu.h"

/* a growing block of page-locked host memory */
struct raw_pin_t {
	uint8_t *p; size_t n, cap;
	raw_pin_t() : p(0), n(0), cap(0) {}
	~raw_pin_t() { ssg_host_free(p); }
	raw_pin_t(const raw_pin_t&) = delete; raw_pin_t &operator=(const raw_pin_t&) = delete;
	bool reserve(size_t want)
	{
		if (want <= cap) return true;
		const size_t c = std::max(want, cap + cap / 2);
		uint8_t *q = (uint8_t*)ssg_host_alloc(c);
		if (!q) return false;
		if (n) memcpy(q, p, n);
		ssg_host_free(p); p = q; cap = c;
		return true;
	}
	void release() { ssg_host_free(p); p = 0; n = cap = 0; }
	bool append(const void *s, size_t k) { if (!reserve(n + k)) return false; memcpy(p + n, s, k); n += k; return true; }
};

/* one input scanned for plain records: a plain regular file by the several-thread scanner, anything else (gzip, bgzip, a pipe) by the one-thread
 * scanner behind fastq.h's decoder threads */
struct raw_src_t {
	std::unique_ptr<rs_pscan_t> p; std::unique_ptr<rs_scan_t> s;
	gzFile fp; std::unique_ptr<fq_stream_t> ks;      /* stream mode: the decoder this scanner reads from (the fall-back parser goes on from it) */
	std::string path; int fd;                      /* file mode: the fall-back parser reads the same file from an offset */
	/* bytes handed out by ptr() and not yet copied: one run */
	const unsigned char *run; size_t run_len; raw_pin_t *run_dst; bool oom;
	raw_src_t(const char *f, gzFile opened) : fp(opened), path(f), fd(-1), run(0), run_len(0), run_dst(0), oom(false)
	{
		if (rs_plain_regular(f)) { p.reset(new rs_pscan_t(f)); p->before_move = [this]() { flush(); }; if (p->slow) p->slow->before_move = p->before_move; fd = open(f, O_RDONLY); }
		else {
			ks.reset(new fq_stream_t(fp, f));
			fq_stream_t *k = ks.get();
			s.reset(new rs_scan_t([k](unsigned char *d, size_t cap) -> long {
				if (k->begin >= k->end && !k->fill()) return 0;
				const size_t n = std::min(cap, (size_t)(k->end - k->begin));
				memcpy(d, k->buf.data() + k->begin, n); k->begin += (int)n;
				return (long)n; }));
			s->before_move = [this]() { flush(); };
		}
	}
	~raw_src_t() { ks.reset(); if (fd >= 0) close(fd); }
	bool io_bad() const { return ks && ks->had_io_err(); }
	int next(size_t *r0, size_t *r1, size_t *len) { return p ? p->next(r0, r1, len) : s->next(r0, r1, len); }
	size_t at() const { return p ? p->at : s->at; }
	const std::string &why() const { return p ? p->why : s->why; }
	const unsigned char *ptr(size_t off) const { return p ? p->ptr(off) : s->ptr(off); }
	void flush() { if (run_len && run_dst && !run_dst->append(run, run_len)) oom = true; run = 0; run_len = 0; }
	/* record [r0, r1) (just returned by next()) goes to dst; returns its offset there */
	size_t take(size_t r0, size_t r1, raw_pin_t *dst)
	{
		const unsigned char *q = ptr(r0);
		if (run_len && (dst != run_dst || q != run + run_len)) flush();
		if (!run_len) { run = q; run_dst = dst; }
		const size_t o = dst->n + run_len;
		run_len += r1 - r0;
		return o;
	}
	/* what the scanner has read beyond `from` and not handed out (stream mode): the fall-back parser starts with these bytes */
	void leftover(size_t from, std::vector<unsigned char> &out)
	{
		if (!s) return;
		if (from < s->base + s->fill) out.insert(out.end(), s->ptr(from), s->ptr(from) + (s->base + s->fill - from));
		if (ks->begin < ks->end) { out.insert(out.end(), ks->buf.data() + ks->begin, ks->buf.data() + ks->end); ks->begin = ks->end; }   /* ... and what the decoder's current chunk still holds */
	}
};

/* one device call's worth of input as text */
struct raw_call_t {
	raw_pin_t txt[2];                      /* the records of the first (or only) input, of the second */
	std::vector<int64_t> rec_off;          /* per read, in read order, over txt[0] ++ txt[1] */
	std::vector<int32_t> pair_batch; int n_batches;
	raw_call_t() : n_batches(0) {}
	int n_pairs() const { return (int)(rec_off.size() / 2); }
};

/* Forms device calls from the scanners.  next_call(): 1 = a call (possibly the last), 0 = the input has ended (nothing in *c), -1 = failure (msg).
 * After a call with `fell_back` set nothing more comes from here: the parser takes over at the start of the first upstream batch that is not
 * in a call -- resume_off[i] in file mode, resume_mem[i] (+ the decoder) in stream mode -- with pairs_done pairs behind it. */
struct raw_feed_t {
	std::unique_ptr<raw_src_t> A, B; int64_t chunk; size_t max_pairs;
	bool eof, fell_back; std::string msg, why; uint64_t pairs_done;
	size_t resume_off[2]; std::vector<unsigned char> resume_mem[2];
	raw_feed_t(const char *f1, gzFile fp1, const char *f2, gzFile fp2, int64_t chunk_, size_t max_pairs_) : A(new raw_src_t(f1, fp1)), B(f2 ? new raw_src_t(f2, fp2) : 0), chunk(chunk_), max_pairs(max_pairs_),
		eof(false), fell_back(false), pairs_done(0) { resume_off[0] = resume_off[1] = 0; }
	int next_call(raw_call_t *c)
	{
		if (eof || fell_back) return 0;
		size_t est = 0;
		while (!eof && !fell_back && (size_t)c->n_pairs() < max_pairs) {   /* upstream bseq_read: one batch */
			int64_t bases = 0; const size_t n0 = c->rec_off.size(), t0[2] = { c->txt[0].n + (A->run_dst == &c->txt[0] ? A->run_len : 0), c->txt[1].n + (B && B->run_dst == &c->txt[1] ? B->run_len : 0) };
			const size_t b_at[2] = { A->at(), B ? B->at() : 0 };
			for (;;) {
				size_t r0, r1, l0, m0, m1, l1;
				int rc = A->next(&r0, &r1, &l0);
				if (rc == 0) { eof = true; break; }
				raw_src_t &S2 = B ? *B : *A;
				const raw_src_t *refused = rc < 0 ? A.get() : &S2;
				if (rc > 0) {
					const size_t o0 = A->take(r0, r1, &c->txt[0]);
					rc = S2.next(&m0, &m1, &l1);
					if (rc == 0) {   /* upstream bseq_read / main_mem: the complete pairs read so far are aligned and printed, the odd read is dropped */
						fprintf(stderr, B ? "[W::bseq_read] the 2nd file has fewer sequences.\n" : "[W::main_mem] odd number of reads in the PE mode; last read dropped\n");
						A->flush(); c->txt[0].n = o0;
						eof = true; break;
					}
					if (rc > 0) {
						if (!est) { est = ((r1 - r0) + (m1 - m0)) * max_pairs; est += est / 16; if (!c->txt[0].reserve(B ? est / 2 : est) || (B && !c->txt[1].reserve(est / 2))) { msg = "out of page-locked memory"; return -1; } }
						const size_t o1 = S2.take(m0, m1, &c->txt[B ? 1 : 0]);
						c->rec_off.push_back((int64_t)o0); c->rec_off.push_back((int64_t)o1);   /* the second input's offsets are shifted by the first's bytes when the call is complete */
						bases += (int64_t)l0 + (int64_t)l1;
						if (bases >= chunk) break;
						continue;
					}
				}
				/* not a plain record: this upstream batch, and everything after it, is the parser's */
				why = refused->path + " has " + refused->why();
				A->flush(); if (B) B->flush();
				fell_back = true;
				for (int i = 0; i < (B ? 2 : 1); ++i) {
					raw_src_t &S = i ? *B : *A;
					if (S.p) resume_off[i] = b_at[i];
					else { resume_mem[i].assign(c->txt[i].p + t0[i], c->txt[i].p + c->txt[i].n); S.leftover(S.at(), resume_mem[i]); }
					c->txt[i].n = t0[i];
				}
				c->rec_off.resize(n0);
				break;
			}
			if (fell_back) break;
			if (A->io_bad() || (B && B->io_bad())) { msg = "the compressed input is damaged"; return -1; }
			if (c->rec_off.size() > n0) { for (size_t p = n0 / 2; p < c->rec_off.size() / 2; ++p) c->pair_batch.push_back(c->n_batches); ++c->n_batches; }
		}
		A->flush(); if (B) B->flush();
		A->run_dst = 0; if (B) B->run_dst = 0;
		if (A->oom || (B && B->oom)) { msg = "out of page-locked memory"; return -1; }
		if (B) for (size_t r = 1; r < c->rec_off.size(); r += 2) c->rec_off[r] += (int64_t)c->txt[0].n;
		pairs_done += (uint64_t)c->n_pairs();
		return c->n_pairs() > 0 ? 1 : 0;
	}
};
#endif
