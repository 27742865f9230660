/*
 * samblaster_main.cpp -- the `samblaster` executable speedseq.config names (reference
 * bin/speedseq.config:14, invoked at bin/speedseq:439,469):
 *   samblaster [--excludeDups] --addMateTags --maxSplitCount INT --minNonOverlap INT
 *              --splitterFile PATH --discordantFile PATH      (SAM on stdin -> SAM on stdout)
 * Host side of the drop-in boundary.  The text is scanned in place (no per-line allocation: only the six leading
 * fields are read, the rest of a line is skipped by memchr), every decision -- duplicate (a14), mate tags' source
 * line (a15), discordant (a16) and splitter (a17) membership -- is taken on the MI355X by ssg_sbl_process over the
 * numeric view of a chunk of blocks, whose duplicate set persists in HBM over the whole stream; the writer then
 * patches FLAG / appends MC,MQ / copies lines to the side streams.  The two side paths are FIFOs in the reference
 * script: they are opened before the first record is read and always closed, even when empty.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <errno.h>
#include <fcntl.h>
#include <string>
#include <vector>
#include <unordered_map>
#include <memory>
#include <thread>
#include <chrono>
#include <atomic>
#include <algorithm>
#include <functional>
#include "../../include/ssgpu.h"
#include "fastq.h"   /* chan_t */
#include "fused.h"
#include "ranks.h"
#include <poll.h>
#include <map>
#include <condition_variable>

static inline void parallel_ranges(int n_threads, size_t n, const std::function<void(size_t, size_t)> &fn)
{
	if (n_threads < 1) n_threads = 1;
	if ((size_t)n_threads > n) n_threads = n ? (int)n : 1;
	if (n_threads == 1) { fn(0, n); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t) th.emplace_back([&, t]() { fn(n * (size_t)t / (size_t)n_threads, n * (size_t)(t + 1) / (size_t)n_threads); });
	for (auto &x : th) x.join();
}

struct out_t {           /* buffered writer on a file descriptor */
	int fd; std::vector<char> b; size_t n;
	explicit out_t(int fd_) : fd(fd_), b(fd_ < 0 ? (size_t)1 << 16 : (size_t)4 << 20), n(0) {}
	void flush() { size_t o = 0; while (o < n) { ssize_t w = write(fd, b.data() + o, n - o); if (w < 0) { if (errno == EINTR) continue; perror("[samblaster] write"); exit(1); } o += (size_t)w; } n = 0; }
	/* fd < 0: everything stays in memory (rank mode: a batch's side-stream lines travel to rank 0) */
	inline void put(const char *p, size_t l) { if (n + l > b.size()) { if (fd < 0) b.resize(std::max(2 * b.size(), n + l)); else { flush(); if (l > b.size()) b.resize(l * 2); } } memcpy(b.data() + n, p, l); n += l; }
	inline void putc(char c) { if (n == b.size()) { if (fd < 0) b.resize(2 * b.size()); else flush(); } b[n++] = c; }
	inline void puti(int v) { char t[16]; int k = 0; if (v == 0) t[k++] = '0'; unsigned u = (unsigned)v; char r[16]; int m = 0; while (u) { r[m++] = (char)('0' + u % 10); u /= 10; } while (m) t[k++] = r[--m]; put(t, (size_t)k); }
};

struct lrec_t {          /* one SAM line inside the chunk buffer */
	size_t off; uint32_t len;            /* line without '\n' */
	uint32_t qn_len, flag_end, cig_off, cig_len, mq_off, mq_len, opt_off;   /* relative to off; opt_off = 0: no optional fields */
};

static inline int parse_uint(const char *p, const char *e) { int v = 0; bool neg = p < e && *p == '-'; if (neg) ++p; while (p < e && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0'); return neg ? -v : v; }

static bool has_tag(const char *p, const char *e, const char *tag)
{	/* optional fields p..e: does one start with tag (5 chars)? */
	while (p < e) {
		if (e - p >= 5 && memcmp(p, tag, 5) == 0) return true;
		const char *t = (const char*)memchr(p, '\t', (size_t)(e - p));
		if (!t) break;
		p = t + 1;
	}
	return false;
}

/* the offsets `emit` needs, from one SAM line (no '\n'); false when the line has fewer than 11 fields */
static bool scan_fields(const char *s, size_t len, size_t off, lrec_t &r, const char **f /* [7] field starts */, const char **opt_out)
{
	const char *e = s + len; int nf = 0;
	f[nf++] = s;
	for (const char *p = s; nf < 7; ) { const char *t = (const char*)memchr(p, '\t', (size_t)(e - p)); if (!t) break; f[nf++] = t + 1; p = t + 1; }
	if (nf < 7) return false;
	const char *p = f[6]; int more = 0;                   /* fields 7..11 are skipped, not read */
	const char *opt = 0;
	while (more < 5) { const char *t = (const char*)memchr(p, '\t', (size_t)(e - p)); if (!t) break; p = t + 1; ++more; }
	if (more < 4) return false;                           /* fewer than 11 mandatory fields */
	if (more == 5) opt = p;
	r.off = off; r.len = (uint32_t)len; r.qn_len = (uint32_t)(f[1] - 1 - s); r.flag_end = (uint32_t)(f[2] - 1 - s);
	r.mq_off = (uint32_t)(f[4] - s); r.mq_len = (uint32_t)(f[5] - 1 - f[4]); r.cig_off = (uint32_t)(f[5] - s); r.cig_len = (uint32_t)(f[6] - 1 - f[5]);
	r.opt_off = opt ? (uint32_t)(opt - s) : 0;
	if (opt_out) *opt_out = opt;
	return true;
}

/* ---------------- fused mode (fused.h): BAM records in, BAM records out, side streams as text ---------------- */
#include <zlib.h>
struct bam_view_t {                     /* fields of one BAM record (after block_size) */
	const uint8_t *p; uint32_t bs;
	int32_t tid() const { int32_t v; memcpy(&v, p, 4); return v; }
	int32_t pos() const { int32_t v; memcpy(&v, p + 4, 4); return v; }
	uint32_t bmn() const { uint32_t v; memcpy(&v, p + 8, 4); return v; }
	uint32_t fnc() const { uint32_t v; memcpy(&v, p + 12, 4); return v; }
	int32_t l_seq() const { int32_t v; memcpy(&v, p + 16, 4); return v; }
	uint32_t l_qname() const { return bmn() & 0xff; }
	uint32_t n_cigar() const { return fnc() & 0xffff; }
	uint32_t flag() const { return fnc() >> 16; }
	uint32_t mapq() const { return (bmn() >> 8) & 0xff; }
	const char *qname() const { return (const char*)p + 32; }
	const uint8_t *cigar() const { return p + 32 + l_qname(); }
	const uint8_t *aux() const { return cigar() + 4 * n_cigar() + ((l_seq() + 1) >> 1) + l_seq(); }
	const uint8_t *end() const { return p + bs; }
};
static bool bam_has_tag(const bam_view_t &v, char a, char b)
{	/* walk of the aux fields (htslib sam.c skip_aux) */
	const uint8_t *s = v.aux(), *e = v.end();
	while (s + 3 <= e) {
		if (s[0] == (uint8_t)a && s[1] == (uint8_t)b) return true;
		const uint8_t t = s[2]; s += 3;
		switch (t) {
			case 'A': case 'c': case 'C': s += 1; break;
			case 's': case 'S': s += 2; break;
			case 'i': case 'I': case 'f': s += 4; break;
			case 'd': s += 8; break;
			case 'Z': case 'H': while (s < e && *s) ++s; ++s; break;
			case 'B': { if (s + 5 > e) return false; const uint8_t st = s[0]; uint32_t n; memcpy(&n, s + 1, 4); const int sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; s += 5 + (size_t)n * sz; break; }
			default: return false;
		}
	}
	return false;
}

/* Rank mode (ranks.h): the duplicate set is SHARDED over the ranks by signature -- every rank's samblaster serves the slice of the table whose signatures
 * hash to it (its own table in its own GPU's HBM), and every rank is a client of all of them: per batch it sends each server the primary ends of the
 * pairs that server owns (an empty slice too: the servers advance batch by batch) and puts the verdicts that come back into pair order; later it sends
 * the batch's side-stream lines to rank 0's server, which owns both side streams.  Every server decides the batches in input order b = 0, 1, 2, ...
 * (batch b comes from rank b mod N): equal signatures always meet in the same server, in the order one samblaster would have seen them in -- first seen
 * wins over all ranks exactly as in one pipeline, with N tables deciding side by side instead of rank 0's alone (SSG_RANKS_SHARD=0: everything to rank 0). */
static inline uint64_t sbl_mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
/* which server owns a pair: a function of its 5'-unclipped signature (the fields csrc/k_misc.h ssg_k_sig keys the table on: upstream samblaster's) and nothing else */
static inline uint64_t sbl_route(const ssg_sbl_end_t *e, uint64_t p)
{
	uint64_t k[2][3]; int m[2];
	for (int i = 0; i < 2; ++i) {
		const ssg_sbl_end_t &x = e[i];
		m[i] = !(x.flag & 0x4) && x.seq >= 0;
		const uint64_t strand = (x.flag & 0x10) ? 1 : 0;
		const int64_t q = strand ? (int64_t)x.pos + x.ralen - 1 + x.rclip : (int64_t)x.pos - x.lclip;
		k[i][0] = (uint64_t)(uint32_t)x.seq; k[i][1] = (uint64_t)(q + (1LL << 31)); k[i][2] = strand;
	}
	uint64_t k0, k1, k2;
	if (m[0] && m[1]) {
		const bool swap = k[0][0] > k[1][0] || (k[0][0] == k[1][0] && (k[0][1] > k[1][1] || (k[0][1] == k[1][1] && k[0][2] > k[1][2])));
		const uint64_t *lo = swap ? k[1] : k[0], *hi = swap ? k[0] : k[1];
		k0 = lo[0] << 32 | hi[0]; k1 = lo[1] << 1 | lo[2]; k2 = hi[1] << 1 | hi[2];
	} else if (m[0] || m[1]) { const uint64_t *a = m[0] ? k[0] : k[1]; k0 = a[0]; k1 = a[1] << 1 | a[2]; k2 = 0; }
	else return p;   /* never a duplicate: anywhere */
	return sbl_mix64(k0 ^ sbl_mix64(k1 ^ sbl_mix64(k2))) >> 7;
}
struct sbl_server_t {
	int world, lfd; ssg_sbl_state_t *st; out_t *spl, *disc;
	std::vector<int> cfd; std::vector<std::thread> readers; std::thread decider;
	std::mutex mu; std::condition_variable cv;
	struct ends_t { int rank; std::vector<uint8_t> p; };
	std::map<uint64_t, ends_t> ends; std::map<uint64_t, std::vector<uint8_t> > side; std::vector<char> done; int failed;
	sbl_server_t(int w, ssg_sbl_state_t *st_, out_t *s, out_t *d) : world(w), lfd(-1), st(st_), spl(s), disc(d), cfd((size_t)w, -1), done((size_t)w, 0), failed(0) {}
	bool start(const std::string &path)
	{
		lfd = rk_listen(path, world);
		if (lfd < 0) return false;
		decider = std::thread([this]() {
			for (int k = 0; k < world; ++k) {     /* every rank says hello with an empty DONE-typed header carrying its rank in `b` = ~0 */
				int fd = -1;
				for (const double t0 = rk_now(); fd < 0; ) {   /* a rank whose pipeline died before its samblaster got here never connects */
					struct pollfd pf; pf.fd = lfd; pf.events = POLLIN; pf.revents = 0;
					if (poll(&pf, 1, 200) > 0) { fd = accept(lfd, 0, 0); break; }
					if (rk_someone_failed() || rk_now() - t0 > rk_timeout()) break;
				}
				rk_hdr_t h; std::vector<uint8_t> pl;
				if (fd < 0 || !rk_recv(fd, h, pl) || h.rank >= (uint32_t)world || cfd[h.rank] >= 0) { fprintf(stderr, "[samblaster] rank mode: a client did not introduce itself\n"); std::lock_guard<std::mutex> l(mu); failed = 1; cv.notify_all(); return; }
				cfd[h.rank] = fd;
			}
			for (int r = 0; r < world; ++r) readers.emplace_back([this, r]() {
				for (;;) {
					rk_hdr_t h; std::vector<uint8_t> pl;
					if (!rk_recv(cfd[(size_t)r], h, pl)) { std::lock_guard<std::mutex> l(mu); if (!done[(size_t)r]) failed = 1; cv.notify_all(); return; }
					std::lock_guard<std::mutex> l(mu);
					if (h.type == RK_ENDS) { ends_t e; e.rank = r; e.p.swap(pl); ends[h.b] = std::move(e); }
					else if (h.type == RK_SIDE) side[h.b].swap(pl);
					else if (h.type == RK_DONE) { done[(size_t)r] = 1; cv.notify_all(); return; }
					cv.notify_all();
				}
			});
			uint64_t next_dup = 0, next_side = 0; std::vector<uint8_t> dup;
			for (;;) {
				ends_t e; std::vector<uint8_t> sd; int what = 0;
				{	std::unique_lock<std::mutex> l(mu);
					/* also woken when the rank that owns the next batch has said it is done while later batches wait here (its pipeline lost one), and
					 * twice a second to look for a rank that gave up elsewhere (ranks.h rk_mark_failed) */
					auto ready = [&] { bool all = true; for (char d : done) all = all && d;
						return failed || ends.count(next_dup) || side.count(next_side) || all || (done[(size_t)(next_dup % (uint64_t)world)] && !ends.empty()) || (done[(size_t)(next_side % (uint64_t)world)] && !side.empty()); };
					while (!ready()) { cv.wait_for(l, std::chrono::milliseconds(500)); if (!ready() && rk_someone_failed()) { failed = 1; break; } }
					if (failed) break;
					if (!ends.count(next_dup) && !side.count(next_side)) { bool all = true; for (char d : done) all = all && d; if (!all || !ends.empty() || !side.empty()) { fprintf(stderr, "[samblaster] rank mode: a batch is missing\n"); failed = 1; break; } }
					if (ends.count(next_dup)) { e = std::move(ends[next_dup]); ends.erase(next_dup); what = 1; }
					else if (side.count(next_side)) { sd.swap(side[next_side]); side.erase(next_side); what = 2; }
					else { if (!ends.empty() || !side.empty()) { fprintf(stderr, "[samblaster] rank mode: a batch is missing\n"); failed = 1; } break; }
				}
				if (what == 1) {
					const long n = (long)(e.p.size() / (2 * sizeof(ssg_sbl_end_t)));
					dup.assign((size_t)n, 0);
					if (n && ssg_sbl_markdup_stream(st, n, (const ssg_sbl_end_t*)e.p.data(), dup.data())) { fprintf(stderr, "[samblaster] %s\n", ssg_last_error()); std::lock_guard<std::mutex> l(mu); failed = 1; break; }
					if (!rk_send(cfd[(size_t)e.rank], RK_DUP, 0, next_dup, dup.data(), dup.size())) { std::lock_guard<std::mutex> l(mu); failed = 1; break; }
					++next_dup;
				} else {   /* payload: u64 bytes of splitter text, then the two texts */
					uint64_t ns = 0; if (sd.size() >= 8) memcpy(&ns, sd.data(), 8);
					if (sd.size() < 8 || ns > sd.size() - 8) { std::lock_guard<std::mutex> l(mu); failed = 1; break; }
					if (spl && ns) spl->put((const char*)sd.data() + 8, (size_t)ns);
					if (disc && sd.size() - 8 - ns) disc->put((const char*)sd.data() + 8 + ns, sd.size() - 8 - (size_t)ns);
					++next_side;
				}
			}
			if (failed) for (int fd : cfd) if (fd >= 0) shutdown(fd, SHUT_RDWR);   /* waiting clients get an error instead of silence */
		});
		return true;
	}
	int finish()
	{
		if (decider.joinable()) decider.join();
		for (std::thread &t : readers) if (t.joinable()) t.join();
		for (int fd : cfd) if (fd >= 0) close(fd);
		if (lfd >= 0) close(lfd);
		return failed;
	}
};

static int fused_main(const ssg_sbl_opt_t &o, out_t *spl, out_t *disc, FILE *splf, FILE *discf, const char *pg)
{
	const int world = rk_world(), rank = rk_rank();
	if (!rk_check("samblaster")) return 1;
	ssg_sbl_state_t *st = ssg_sbl_state_new();
	unsigned long long n_pairs = 0, n_dups = 0, n_disc = 0, n_spl = 0;
	int threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
	{ const char *e = getenv("SSG_SBL_THREADS"); if (e && atoi(e) > 0) threads = atoi(e); }   /* (not cut to the cgroup's cores as the sort's pool is: its bursts are short, and 16 threads instead of 32 lengthened the end of an 8 M-pair run by 0.3 s) */
	std::unique_ptr<sbl_server_t> srv; std::vector<int> cfds; std::mutex c_mu;   /* rank mode: this rank's slice of the duplicate set (rank 0's: the side streams too) and its connections to every rank's */
	const bool shard = !(getenv("SSG_RANKS_SHARD") && !strcmp(getenv("SSG_RANKS_SHARD"), "0"));
	if (world > 1) {
		srv.reset(new sbl_server_t(world, st, rank == 0 ? spl : 0, rank == 0 ? disc : 0));
		if (!srv->start(rk_dir() + "/sbl." + std::to_string(rank) + ".sock")) return 1;
		cfds.assign((size_t)world, -1);
		for (int q = 0; q < world; ++q) {
			cfds[(size_t)q] = rk_connect(rk_dir() + "/sbl." + std::to_string(q) + ".sock");
			if (cfds[(size_t)q] < 0 || !rk_send(cfds[(size_t)q], 0, rank, 0, 0, 0)) { fprintf(stderr, "[samblaster] rank mode: cannot reach rank %d's samblaster\n", q); return 1; }
		}
	}
	const int cfd = world > 1 ? cfds[0] : -1;   /* the side streams' owner */
	uint64_t n_batch = 0;                                     /* BATCH frames seen: frame k of this rank is batch rank + k * world of the input */
	if (!fu_write_full(1, FU_MAGIC, 8)) { perror("[samblaster] write"); return 1; }
	bool got_header = false, ended = false;
	/* frames are read by a thread of their own so that the next batch arrives while this one is decided and written */
	struct frame_t { fu_frame_t h; fu_buf_t b; };
	chan_t<std::unique_ptr<frame_t> > ch(2); std::atomic<int> rd_fail(0);
	/* heap frame buffers go back to the reader when a batch is done: hundreds of MB of warm memory instead of fresh pages per frame */
	std::mutex pool_mu; std::vector<std::unique_ptr<frame_t> > pool;
	std::thread reader([&]() {
		for (;;) {
			std::unique_ptr<frame_t> F;
			{ std::lock_guard<std::mutex> l(pool_mu); if (!pool.empty()) { F = std::move(pool.back()); pool.pop_back(); } }
			if (!F) F.reset(new frame_t());
			if (!fu_read_frame(0, F->h, F->b)) { rd_fail = 1; fu_discard_rest(0); break; }
			const bool end = F->h.type == FU_END;
			ch.push(std::move(F));
			if (end) break;
		}
		ch.close();
	});
	/* Two stages, a batch each: this thread takes a frame, makes samblaster's numeric view of its lines and has the device decide (the duplicate
	 * table makes that stage sequential by nature); a second thread rebuilds the records, writes the main frame and the side streams of the
	 * batch before, in order.  (One thread doing both was 3.6 s of work per 8 M pairs on the MI355X box, as much as `bwa mem` needed for them.) */
	struct work_t {
		std::unique_ptr<frame_t> F; fu_batch_t bh; const fu_cand_t *cand; const char *text; const uint8_t *bam; size_t nr, n_blocks; uint64_t b;
		std::vector<uint64_t> rec_off; std::vector<ssg_sbl_line_t> lines; std::vector<uint8_t> newblk, bits; std::vector<int64_t> blk_off, mate;
	};
	std::vector<std::pair<const char*, uint32_t> > ltext;
	std::vector<std::vector<uint8_t> > outb;                  /* per-thread output of a batch, kept across batches */
	std::unique_ptr<frame_t> F;
	std::atomic<int> rc(0);
	chan_t<std::unique_ptr<work_t> > to_b(1);
	std::mutex wpool_mu; std::vector<std::unique_ptr<work_t> > wpool;
	double tm[6] = { 0, 0, 0, 0, 0, 0 };                     /* wait for a frame, numeric view, decisions, records rebuilt, main stream written, side streams */
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tmb[4] = { 0, 0, 0, 0 };                          /* second stage: wait for a decided batch, records rebuilt, main stream written, side streams */
	/* ... and a third thread writes the batch's side streams (the FIFOs' readers -- gawk, `sambamba view -S` -- take their time: 1.3 s of an 8 M-pair run were spent in
	 * those writes by the thread that also had the next batch's records to rebuild, at the pace `bwa mem' now delivers them) */
	chan_t<std::unique_ptr<work_t> > to_c(1);
	std::thread stage_c([&]() {
		std::unique_ptr<work_t> W;
		for (;;) {
			if (!to_c.pop(W)) break;
			if (!rc) {
			std::unique_ptr<frame_t> &F = W->F; const fu_batch_t &bh = W->bh; const fu_cand_t *cand = W->cand; const char *text = W->text;
			const size_t nr = W->nr, n_blocks = W->n_blocks;
			std::vector<ssg_sbl_line_t> &lines = W->lines; std::vector<uint8_t> &bits = W->bits; std::vector<int64_t> &blk_off = W->blk_off, &mate = W->mate;
			double t0 = now();
			/* side streams, from the text bwa attached for the pairs that can qualify */
			ltext.assign(nr, std::pair<const char*, uint32_t>((const char*)0, 0u));
			for (uint64_t c = 0; c < bh.n_cand; ++c) {
				const char *p = text + cand[c].text_off, *e = text + (c + 1 < bh.n_cand ? cand[c + 1].text_off : bh.text_bytes);
				for (uint64_t k = 0; k < cand[c].n_rec && p < e; ++k) {
					const char *nl = (const char*)memchr(p, '\n', (size_t)(e - p)); if (!nl) nl = e;
					if (cand[c].first_rec + k < nr) ltext[(size_t)(cand[c].first_rec + k)] = std::make_pair(p, (uint32_t)(nl - p));
					p = nl < e ? nl + 1 : e;
				}
			}
			auto side = [&](out_t &w, size_t i, int flag, bool patch, const char *suffix) -> bool {
				if (!ltext[i].first) return false;
				lrec_t r, mr; const char *f[12]; const char *opt = 0;
				if (!scan_fields(ltext[i].first, ltext[i].second, 0, r, f, &opt)) return false;
				const lrec_t *m = 0; bool add_mc = false, add_mq = false; const char *mbase = 0;
				if (o.add_mate_tags && mate[i] >= 0) {
					const size_t mi = (size_t)mate[i]; const char *g[12];
					if (!ltext[mi].first || !scan_fields(ltext[mi].first, ltext[mi].second, 0, mr, g, 0)) return false;
					m = &mr; mbase = ltext[mi].first;
					const char *oe = ltext[i].first + ltext[i].second;
					add_mc = !(opt && has_tag(opt, oe, "MC:Z:")); add_mq = !(opt && has_tag(opt, oe, "MQ:i:"));
				}
				/* emit_line addresses the line and its mate through one base pointer: write the two parts separately */
				const char *s0 = ltext[i].first;
				if (!patch && !suffix) w.put(s0, r.len);
				else { w.put(s0, r.qn_len); if (suffix) w.put(suffix, 2); w.putc('\t'); w.puti(flag); w.put(s0 + r.flag_end, r.len - r.flag_end); }
				if (m) { if (add_mc) { w.put("\tMC:Z:", 6); w.put(mbase + m->cig_off, m->cig_len); } if (add_mq) { w.put("\tMQ:i:", 6); w.put(mbase + m->mq_off, m->mq_len); } }
				w.putc('\n');
				return true;
			};
			/* every block is looked at for the counters, few (a few per cent) put lines on a side stream: the look is done by the pool, the lines are
			 * written in block order by this thread */
			struct sideblk_t { int64_t b, d1, d2; bool dup, split; };
			const int TS = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n_blocks / 16384 + 1));
			std::vector<std::vector<sideblk_t> > found((size_t)TS); std::vector<unsigned long long> c_pairs((size_t)TS, 0), c_dups((size_t)TS, 0);
			parallel_ranges(TS, (size_t)TS, [&](size_t ta, size_t tb) {
				for (size_t t = ta; t < tb; ++t) {
					for (size_t b = n_blocks * t / (size_t)TS, e = n_blocks * (t + 1) / (size_t)TS; b < e; ++b) {
						bool paired = false, dup = false, split = false; int64_t d1 = -1, d2 = -1;
						for (int64_t i = blk_off[b]; i < blk_off[b + 1]; ++i) {
							const int bt = bits[(size_t)i];
							if (mate[(size_t)i] >= 0) paired = true;
							if (bt & SSG_SBL_DUP) dup = true;
							if (bt & SSG_SBL_DISC) { if (lines[(size_t)i].flag & 0x40) d1 = i; else d2 = i; }
							if (bt & SSG_SBL_SPLIT) split = true;
						}
						if (paired) { ++c_pairs[t]; if (dup) ++c_dups[t]; }
						if ((disc && d1 >= 0 && d2 >= 0) || (spl && split)) { sideblk_t k; k.b = (int64_t)b; k.d1 = d1; k.d2 = d2; k.dup = dup; k.split = split; found[t].push_back(k); }
					}
				}
			});
			for (int t = 0; t < TS; ++t) { n_pairs += c_pairs[(size_t)t]; n_dups += c_dups[(size_t)t]; }
			out_t mem_spl(-1), mem_disc(-1);                      /* rank mode: the batch's lines are collected and sent to rank 0, which writes every rank's in batch order */
			out_t *const spl_w = world > 1 ? &mem_spl : spl, *const disc_w = world > 1 ? &mem_disc : disc;
			for (int t = 0; t < TS && !rc; ++t) for (const sideblk_t &k : found[(size_t)t]) {
				const bool dup = k.dup; const int64_t d1 = k.d1, d2 = k.d2; const size_t b = (size_t)k.b;
				if (disc && d1 >= 0 && d2 >= 0) {
					for (int64_t i : { d1, d2 }) if (!side(*disc_w, (size_t)i, lines[(size_t)i].flag | (dup ? 0x400 : 0), dup, 0)) { fprintf(stderr, "[samblaster] fused stream: a discordant line came without its text\n"); rc = 1; }
					++n_disc;
				}
				if (spl && k.split) for (int64_t i = blk_off[b]; i < blk_off[b + 1]; ++i) if (bits[(size_t)i] & SSG_SBL_SPLIT) {
					if (!side(*spl_w, (size_t)i, lines[(size_t)i].flag | (dup ? 0x400 : 0), true, (lines[(size_t)i].flag & 0x40) ? "_1" : "_2")) { fprintf(stderr, "[samblaster] fused stream: a splitter line came without its text\n"); rc = 1; }
					++n_spl;
				}
				if (rc) break;
			}
			if (world > 1 && !rc) {
				const uint64_t ns = (uint64_t)mem_spl.n;
				std::lock_guard<std::mutex> l(c_mu);
				if (!rk_send(cfd, RK_SIDE, rank, W->b, &ns, 8, mem_spl.b.data(), mem_spl.n, mem_disc.b.data(), mem_disc.n)) { fprintf(stderr, "[samblaster] rank mode: rank 0's samblaster is gone\n"); rc = 1; }
			}
			if (F->b.mapped) F->b.reset();                           /* the segment's pages go back to the system now */
			{ std::lock_guard<std::mutex> l(pool_mu); if (pool.size() < 3) pool.push_back(std::move(F)); }
			tmb[3] += now() - t0;
			}
			{ std::lock_guard<std::mutex> l(wpool_mu); if (wpool.size() < 3) wpool.push_back(std::move(W)); }
		}
	});
	std::thread stage_b([&]() {
		std::unique_ptr<work_t> W;
		for (;;) {
			{ const double t0 = now(); const bool got = to_b.pop(W); tmb[0] += now() - t0; if (!got) break; }
			if (rc) continue;                                     /* a failed run: take the batches off the channel, do nothing with them */
			std::unique_ptr<frame_t> &F = W->F; const fu_batch_t &bh = W->bh; const fu_cand_t *cand = W->cand; const char *text = W->text; const uint8_t *bam = W->bam;
			const size_t nr = W->nr, n_blocks = W->n_blocks;
			std::vector<uint64_t> &rec_off = W->rec_off; std::vector<ssg_sbl_line_t> &lines = W->lines; std::vector<uint8_t> &bits = W->bits; std::vector<int64_t> &blk_off = W->blk_off, &mate = W->mate;
			auto view = [&](size_t i) { bam_view_t v; v.p = bam + rec_off[i] + 4; v.bs = (uint32_t)(rec_off[i + 1] - rec_off[i] - 4); return v; };
			double t0 = now();
			do {
			const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n_blocks / 4096 + 1));
			if (outb.size() < (size_t)T) outb.resize((size_t)T);
			for (auto &v : outb) v.clear();
			parallel_ranges(T, (size_t)T, [&](size_t ta, size_t tb) {
				for (size_t t = ta; t < tb; ++t) {
					const size_t b0 = n_blocks * t / (size_t)T, b1 = n_blocks * (t + 1) / (size_t)T;
					std::vector<uint8_t> &ob = outb[t];
					ob.reserve((size_t)((rec_off[(size_t)blk_off[b1]] - rec_off[(size_t)blk_off[b0]]) * 21 / 20) + 4096);
					char cg[16];
					for (size_t i = (size_t)blk_off[b0]; i < (size_t)blk_off[b1]; ++i) {
						const bam_view_t v = view(i);
						const size_t base = ob.size();
						ob.insert(ob.end(), bam + rec_off[i], bam + rec_off[i + 1]);
						if (bits[i] & SSG_SBL_DUP) { uint32_t fnc; memcpy(&fnc, ob.data() + base + 4 + 12, 4); fnc |= 0x400u << 16; memcpy(ob.data() + base + 4 + 12, &fnc, 4); }
						if (o.add_mate_tags && mate[i] >= 0) {
							const bam_view_t m = view((size_t)mate[i]);
							if (!bam_has_tag(v, 'M', 'C')) {
								ob.push_back('M'); ob.push_back('C'); ob.push_back('Z');
								if (!m.n_cigar()) ob.push_back('*');
								for (uint32_t k = 0; k < m.n_cigar(); ++k) {
									uint32_t x; memcpy(&x, m.cigar() + 4 * k, 4); uint32_t len = x >> 4; int n = 0;
									do { cg[n++] = (char)('0' + len % 10); len /= 10; } while (len);
									while (n) ob.push_back((uint8_t)cg[--n]);
									ob.push_back((uint8_t)"MIDNSHP=XB"[x & 0xf]);
								}
								ob.push_back(0);
							}
							if (!bam_has_tag(v, 'M', 'Q')) { ob.push_back('M'); ob.push_back('Q'); ob.push_back('C'); ob.push_back((uint8_t)m.mapq()); }   /* MAPQ <= 255: sam_parse1 types it 'C' */
							const uint32_t nbs = (uint32_t)(ob.size() - base - 4); memcpy(ob.data() + base, &nbs, 4);
						}
					}
				}
			});
			tmb[1] += now() - t0; t0 = now();
			{	uint64_t tot = 0; std::vector<uint64_t> at(outb.size() + 1, 0);
				for (size_t k = 0; k < outb.size(); ++k) { at[k] = tot; tot += outb[k].size(); }
				fu_buf_t seg; std::string seg_path; bool ok;
				if (fu_seg_create((size_t)tot, seg, seg_path)) {       /* the frame as a mapped segment: filled by the threads, only its name travels */
					parallel_ranges((int)outb.size(), outb.size(), [&](size_t a, size_t b) { for (size_t k = a; k < b; ++k) if (!outb[k].empty()) memcpy(seg.p + at[k], outb[k].data(), outb[k].size()); });
					ok = fu_seg_send(1, FU_MAIN, seg, seg_path);
				} else {
					fu_frame_t fh; fh.type = FU_MAIN; fh.zero = 0; fh.len = tot;
					ok = fu_write_full(1, &fh, sizeof(fh));
					for (auto &v : outb) ok = ok && fu_write_full(1, v.data(), v.size());
				}
				if (!ok) { perror("[samblaster] write"); rc = 1; } }
			if (rc) break;
			tmb[2] += now() - t0; t0 = now();
			} while (0);
			if (!rc) { to_c.push(std::move(W)); continue; }   /* the side streams of this batch: the third stage */
			{ std::lock_guard<std::mutex> l(wpool_mu); if (wpool.size() < 3) wpool.push_back(std::move(W)); }
		}
	});
	for (;;) {
		{ const double t0 = now(); const bool got = !rc && ch.pop(F); tm[0] += now() - t0; if (!got) break; }
		if (F->h.type == FU_END) { ended = true; break; }
		if (F->h.type == FU_HEADER) {
			std::string h((const char*)F->b.p, (size_t)F->h.len); h += pg;
			if (!fu_write_frame(1, FU_HEADER, h.data(), h.size())) { perror("[samblaster] write"); rc = 1; break; }
			if (spl) spl->put(h.data(), h.size());
			if (disc) disc->put(h.data(), h.size());
			got_header = true; continue;
		}
		if (F->h.type != FU_BATCH || F->h.len < sizeof(fu_batch_t)) { fprintf(stderr, "[samblaster] unexpected frame in the fused stream\n"); rc = 1; break; }
		std::unique_ptr<work_t> W;
		{ std::lock_guard<std::mutex> l(wpool_mu); if (!wpool.empty()) { W = std::move(wpool.back()); wpool.pop_back(); } }
		if (!W) W.reset(new work_t());
		fu_batch_t &bh = W->bh; memcpy(&bh, F->b.p, sizeof(bh));
		if (sizeof(bh) + bh.n_cand * sizeof(fu_cand_t) + bh.text_bytes + bh.bam_bytes != F->h.len) { fprintf(stderr, "[samblaster] malformed batch frame\n"); rc = 1; break; }
		const fu_cand_t *cand = W->cand = (const fu_cand_t*)(F->b.p + sizeof(bh));
		const char *text = W->text = (const char*)(cand + bh.n_cand);
		const uint8_t *bam = W->bam = (const uint8_t*)text + bh.text_bytes;
		const size_t nr = W->nr = (size_t)bh.n_rec;
		W->b = (uint64_t)rank + n_batch++ * (uint64_t)world;
		if (!nr && world > 1) { fprintf(stderr, "[samblaster] rank mode: an empty batch\n"); rc = 1; break; }
		if (!nr) { std::lock_guard<std::mutex> l(pool_mu); if (pool.size() < 4) pool.push_back(std::move(F)); continue; }
		std::vector<uint64_t> &rec_off = W->rec_off; std::vector<ssg_sbl_line_t> &lines = W->lines; std::vector<uint8_t> &newblk = W->newblk, &bits = W->bits; std::vector<int64_t> &blk_off = W->blk_off, &mate = W->mate;
		double t0 = now();
		rec_off.resize(nr + 1);
		{ uint64_t o2 = 0; size_t i = 0; for (; i < nr && o2 + 4 <= bh.bam_bytes; ++i) { rec_off[i] = o2; uint32_t bs; memcpy(&bs, bam + o2, 4); o2 += 4 + (uint64_t)bs; } rec_off[nr] = o2;
		  if (i != nr || o2 != bh.bam_bytes) { fprintf(stderr, "[samblaster] record count does not match the batch frame\n"); rc = 1; break; } }
		lines.resize(nr); newblk.resize(nr); bits.resize(nr); mate.resize(nr);
		auto view = [&](size_t i) { bam_view_t v; v.p = bam + rec_off[i] + 4; v.bs = (uint32_t)(rec_off[i + 1] - rec_off[i] - 4); return v; };
		/* samblaster's numeric view of every line, from the BAM fields the text would have carried */
		parallel_ranges(threads, nr, [&](size_t a, size_t b) {
			for (size_t i = a; i < b; ++i) {
				const bam_view_t v = view(i); ssg_sbl_line_t n;
				n.flag = (int32_t)v.flag(); n.pos = v.pos() + 1; n.mapq = (int32_t)v.mapq(); n.seq = v.tid() < 0 ? -1 : v.tid();
				n.lclip = n.rclip = n.qalen = n.ralen = 0;
				bool first = true; int rcl = 0; const uint8_t *c = v.cigar();
				for (uint32_t k = 0; k < v.n_cigar(); ++k) {
					uint32_t x; memcpy(&x, c + 4 * k, 4); const int op = (int)(x & 0xf), len = (int)(x >> 4);
					if (op == 4 || op == 5) { if (first) n.lclip += len; rcl += len; }
					else { first = false; rcl = 0; if (op == 0 || op == 7 || op == 8) { n.qalen += len; n.ralen += len; } else if (op == 1) n.qalen += len; else if (op == 2 || op == 3) n.ralen += len; }
				}
				n.rclip = (n.qalen + n.ralen) ? rcl : 0;
				lines[i] = n;
				newblk[i] = (i == 0 || strcmp(view(i - 1).qname(), v.qname()) != 0) ? 1 : 0;
			}
		});
		blk_off.clear();
		for (size_t i = 0; i < nr; ++i) if (newblk[i]) blk_off.push_back((int64_t)i);
		const size_t n_blocks = blk_off.size(); blk_off.push_back((int64_t)nr);
		tm[1] += now() - t0; t0 = now();
		if (world == 1) { if (ssg_sbl_process(st, &o, (long)n_blocks, blk_off.data(), lines.data(), bits.data(), mate.data())) { fprintf(stderr, "[samblaster] %s\n", ssg_last_error()); rc = 1; break; } }
		else {   /* the ends of the blocks' primaries to the owner of the duplicate set, its verdicts back, then the lines' bits */
			std::vector<ssg_sbl_end_t> ends(2 * n_blocks); rk_hdr_t rh; std::vector<uint8_t> dup(n_blocks), part;
			bool ok = ssg_sbl_ends((long)n_blocks, blk_off.data(), lines.data(), ends.data()) == 0;
			std::vector<std::vector<ssg_sbl_end_t> > slice((size_t)world); std::vector<std::vector<uint32_t> > who((size_t)world);
			for (size_t p = 0; ok && p < n_blocks; ++p) {   /* pair order within a slice = pair order within the batch */
				const size_t q = shard ? (size_t)(sbl_route(&ends[2 * p], (uint64_t)p) % (uint64_t)world) : 0;
				slice[q].push_back(ends[2 * p]); slice[q].push_back(ends[2 * p + 1]); who[q].push_back((uint32_t)p);
			}
			if (ok) { std::lock_guard<std::mutex> l(c_mu); for (int q = 0; ok && q < world; ++q) ok = rk_send(cfds[(size_t)q], RK_ENDS, rank, W->b, slice[(size_t)q].data(), slice[(size_t)q].size() * sizeof(ssg_sbl_end_t)); }
			for (int q = 0; ok && q < world; ++q) {
				ok = rk_recv(cfds[(size_t)q], rh, part) && rh.type == RK_DUP && rh.b == W->b && part.size() == who[(size_t)q].size();
				for (size_t k = 0; ok && k < part.size(); ++k) dup[who[(size_t)q][k]] = part[k];
			}
			if (!ok || ssg_sbl_classify(&o, (long)n_blocks, blk_off.data(), lines.data(), dup.data(), bits.data(), mate.data())) {
				fprintf(stderr, "[samblaster] rank mode: batch %llu was not decided (%s)\n", (unsigned long long)W->b, ok ? ssg_last_error() : "another rank's samblaster is gone"); rc = 1; break; }
		}
		tm[2] += now() - t0;
		W->n_blocks = n_blocks; W->F = std::move(F);
		to_b.push(std::move(W));
	}
	to_b.close();
	stage_b.join();
	to_c.close();
	stage_c.join();
	tm[3] = tmb[1]; tm[4] = tmb[2]; tm[5] = tmb[3];
	if (getenv("SSG_SBL_LOG")) fprintf(stderr, "[samblaster] first stage: waiting for frames %.2f s, numeric view %.2f s, decisions %.2f s; second stage: waiting for decided batches %.2f s, records %.2f s, main stream written %.2f s, side streams %.2f s\n", tm[0], tm[1], tm[2], tmb[0], tm[3], tm[4], tm[5]);
	{ std::unique_ptr<frame_t> drop; while (ch.pop(drop)) {} }
	reader.join();
	if (!rc && (rd_fail || !ended)) { fprintf(stderr, "[samblaster] the fused stream ended early\n"); rc = 1; }
	if (!rc && !got_header) { if (spl) spl->put(pg, strlen(pg)); if (disc) disc->put(pg, strlen(pg)); }
	if (!rc && !fu_write_frame(1, FU_END, 0, 0)) rc = 1;
	if (world > 1) {
		if (rc) rk_mark_failed("samblaster");
		{ std::lock_guard<std::mutex> l(c_mu); for (int fd : cfds) if (fd >= 0) (void)rk_send(fd, RK_DONE, rank, 0, 0, 0); }
		if (srv && srv->finish()) { fprintf(stderr, "[samblaster] rank mode: the exchange between the ranks failed\n"); rc = 1; }
		for (int fd : cfds) if (fd >= 0) close(fd);
	}
	if (spl) { spl->flush(); fclose(splf); }
	if (disc) { disc->flush(); fclose(discf); }
	ssg_sbl_state_free(st);
	fprintf(stderr, "[samblaster] pairs=%llu dups=%llu discordant_pairs=%llu splitter_lines=%llu (fused stream: BAM records; decisions on %s)\n", n_pairs, n_dups, n_disc, n_spl, ssg_backend());
	return rc;
}

static int main_(int argc, char **argv);
int main(int argc, char **argv) { ssg_stamp("samblaster", "start"); const int rc = main_(argc, argv); ssg_stamp("samblaster", "end"); return ssg_fast_exit(rc); }
static int main_(int argc, char **argv)
{
	ssg_sbl_opt_t o; ssg_sbl_opt_init(&o);
	const char *spl_path = 0, *disc_path = 0;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--excludeDups")) o.exclude_dups = 1;
		else if (!strcmp(argv[i], "--addMateTags")) o.add_mate_tags = 1;
		else if (!strcmp(argv[i], "--maxSplitCount") && i + 1 < argc) o.max_split_count = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--minNonOverlap") && i + 1 < argc) o.min_non_overlap = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--maxUnmappedBases") && i + 1 < argc) o.max_unmapped_bases = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--minIndelSize") && i + 1 < argc) o.min_indel_size = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--splitterFile") && i + 1 < argc) spl_path = argv[++i];
		else if (!strcmp(argv[i], "--discordantFile") && i + 1 < argc) disc_path = argv[++i];
		else { fprintf(stderr, "[samblaster] unsupported option %s\n", argv[i]); return 1; }
	}
	FILE *splf = spl_path ? fopen(spl_path, "w") : 0, *discf = disc_path ? fopen(disc_path, "w") : 0;
	if ((spl_path && !splf) || (disc_path && !discf)) { fprintf(stderr, "[samblaster] cannot open a side file\n"); return 1; }
	out_t out(1); out_t *spl = splf ? new out_t(fileno(splf)) : 0, *disc = discf ? new out_t(fileno(discf)) : 0;
#ifdef F_SETPIPE_SZ
	(void)fcntl(0, F_SETPIPE_SZ, 1 << 20); (void)fcntl(1, F_SETPIPE_SZ, 1 << 20);
#endif
	const char *pg = "@PG\tID:SAMBLASTER\tVN:0.1.22-ssgpu\tCL:samblaster\n";
	/* the first bytes say whether bwa sent SAM text or the fused stream of BAM records (fused.h) */
	char first[8]; size_t n_first = 0;
	while (n_first < 8) { ssize_t r = read(0, first + n_first, 8 - n_first); if (r < 0) { if (errno == EINTR) continue; perror("[samblaster] read"); return 1; } if (r == 0) break; n_first += (size_t)r; }
	if (n_first == 8 && !memcmp(first, FU_MAGIC, 8)) return fused_main(o, spl, disc, splf, discf, pg);
	ssg_sbl_state_t *st = ssg_sbl_state_new();
	std::unordered_map<std::string, int> seqs;
	size_t CHUNK = 1u << 18;     /* blocks per device call */
	{ const char *e = getenv("SSG_SBL_CHUNK"); if (e && atol(e) > 0) CHUNK = (size_t)atol(e); }
	unsigned long long n_pairs = 0, n_dups = 0, n_disc = 0, n_spl = 0;

	/* a chunk of the stream: text + the in-place view of its complete blocks.  The reader thread fills and parses chunks, this
	 * thread takes the device decisions and writes; the two overlap through a bounded channel. */
	struct chunk_t {
		std::vector<char> buf; size_t have, scan; std::string header;   /* header: @-lines (+ the @PG line) that precede this chunk's records */
		std::vector<lrec_t> L; std::vector<ssg_sbl_line_t> N; std::vector<int64_t> offs;   /* offs: block starts, then the end */
		chunk_t() : buf(64u << 20), have(0), scan(0) {}
	};
	chan_t<std::unique_ptr<chunk_t> > ch(2);
	std::atomic<bool> in_header(true); std::atomic<int> parse_fail(0);   /* written by the reader thread, read by this one */
	std::string last_rname; int last_seq = -1;

	auto parse_line = [&](chunk_t &C, size_t off, size_t len) -> bool {
		const char *s = C.buf.data() + off, *e = s + len, *f[12]; int nf = 0;
		f[nf++] = s;
		for (const char *p = s; nf < 7; ) { const char *t = (const char*)memchr(p, '\t', (size_t)(e - p)); if (!t) break; f[nf++] = t + 1; p = t + 1; }
		if (nf < 7) return false;
		const char *p = f[6]; int more = 0;                   /* fields 7..11 are skipped, not read */
		const char *opt = 0;
		while (more < 5) { const char *t = (const char*)memchr(p, '\t', (size_t)(e - p)); if (!t) break; p = t + 1; ++more; }
		if (more < 4) return false;                           /* fewer than 11 mandatory fields */
		if (more == 5) opt = p;
		lrec_t r; r.off = off; r.len = (uint32_t)len; r.qn_len = (uint32_t)(f[1] - 1 - s); r.flag_end = (uint32_t)(f[2] - 1 - s);
		r.mq_off = (uint32_t)(f[4] - s); r.mq_len = (uint32_t)(f[5] - 1 - f[4]); r.cig_off = (uint32_t)(f[5] - s); r.cig_len = (uint32_t)(f[6] - 1 - f[5]);
		r.opt_off = opt ? (uint32_t)(opt - s) : 0;
		ssg_sbl_line_t n; n.flag = parse_uint(f[1], f[2] - 1); n.pos = parse_uint(f[3], f[4] - 1); n.mapq = parse_uint(f[4], f[5] - 1);
		const size_t rl = (size_t)(f[3] - 1 - f[2]);
		if (rl == 1 && f[2][0] == '*') n.seq = -1;
		else if (rl == last_rname.size() && memcmp(f[2], last_rname.data(), rl) == 0) n.seq = last_seq;
		else { last_rname.assign(f[2], rl); auto it = seqs.find(last_rname); last_seq = n.seq = it == seqs.end() ? -1 : it->second; }
		n.lclip = n.rclip = n.qalen = n.ralen = 0;
		const char *c = f[5], *ce = f[6] - 1;
		if (!(ce - c == 1 && *c == '*')) {
			bool first = true; int rc = 0;
			while (c < ce) {
				int k = 0; while (c < ce && *c >= '0' && *c <= '9') k = k * 10 + (*c++ - '0');
				const char op = c < ce ? *c++ : 0;
				if (op == 'S' || op == 'H') { if (first) n.lclip += k; rc += k; }
				else { first = false; rc = 0; if (op == 'M' || op == '=' || op == 'X') { n.qalen += k; n.ralen += k; } else if (op == 'I') n.qalen += k; else if (op == 'D' || op == 'N') n.ralen += k; }
			}
			n.rclip = (n.qalen + n.ralen) ? rc : 0;
		}
		C.L.push_back(r); C.N.push_back(n);
		return true;
	};

	std::thread reader([&]() {
		std::unique_ptr<chunk_t> C(new chunk_t()); bool eof = false;
		memcpy(C->buf.data(), first, n_first); C->have = n_first;          /* the bytes the format check consumed */
		while (!eof && !parse_fail) {
			if (C->have == C->buf.size()) C->buf.resize(C->buf.size() * 2);
			ssize_t r = read(0, C->buf.data() + C->have, C->buf.size() - C->have);
			if (r < 0) { if (errno == EINTR) continue; perror("[samblaster] read"); parse_fail = 1; break; }
			if (r == 0) { eof = true; if (C->have > C->scan && C->buf[C->have - 1] != '\n') { if (C->have == C->buf.size()) C->buf.resize(C->buf.size() + 1); C->buf[C->have++] = '\n'; } }
			else C->have += (size_t)r;
			while (C->scan < C->have) {   /* parse complete lines */
				const char *nlp = (const char*)memchr(C->buf.data() + C->scan, '\n', C->have - C->scan);
				if (!nlp) break;
				size_t len = (size_t)(nlp - (C->buf.data() + C->scan)), lo = C->scan;
				C->scan += len + 1;
				while (len && C->buf[lo + len - 1] == '\r') --len;
				if (in_header && len && C->buf[lo] == '@') {
					if (len > 3 && !memcmp(C->buf.data() + lo, "@SQ", 3)) {
						std::string ln(C->buf.data() + lo, len); size_t q = ln.find("\tSN:");
						if (q != std::string::npos) { q += 4; size_t e = ln.find('\t', q); std::string nm = ln.substr(q, e == std::string::npos ? std::string::npos : e - q); int id = (int)seqs.size(); seqs.emplace(nm, id); }
					}
					C->header.append(C->buf.data() + lo, len); C->header.push_back('\n');
					continue;
				}
				if (in_header) { in_header = false; C->header.append(pg); }
				if (!len) continue;
				if (!parse_line(*C, lo, len)) { fprintf(stderr, "[samblaster] malformed SAM line\n"); parse_fail = 1; break; }
				const size_t i = C->L.size() - 1;
				if (i == 0 || !(C->L[i - 1].qn_len == C->L[i].qn_len && memcmp(C->buf.data() + C->L[i - 1].off, C->buf.data() + C->L[i].off, C->L[i].qn_len) == 0)) C->offs.push_back((int64_t)i);
			}
			/* blocks complete so far: all but the last (it may continue in the next read) unless EOF */
			const size_t n_open = C->offs.size(), n_done = eof ? n_open : (n_open ? n_open - 1 : 0);
			if (n_done >= CHUNK || eof) {
				const int64_t cut_line = eof ? (int64_t)C->L.size() : C->offs[n_done];
				std::unique_ptr<chunk_t> nx;
				if (!eof) {   /* the open block's lines and the unparsed tail start the next chunk (re-parsed there) */
					nx.reset(new chunk_t());
					const size_t cut_byte = (size_t)cut_line < C->L.size() ? C->L[(size_t)cut_line].off : C->scan;
					const size_t keep = C->have - cut_byte;
					if (keep > nx->buf.size()) nx->buf.resize(keep * 2);
					memcpy(nx->buf.data(), C->buf.data() + cut_byte, keep); nx->have = keep;
				}
				C->L.resize((size_t)cut_line); C->N.resize((size_t)cut_line); C->offs.resize(n_done); C->offs.push_back(cut_line);
				ch.push(std::move(C));
				C = std::move(nx);
			}
		}
		ch.close();
	});

	std::vector<uint8_t> bits; std::vector<int64_t> mate;
	std::unique_ptr<chunk_t> C;
	bool wrote_header = false;
	while (ch.pop(C)) {
		chunk_t &K = *C;
		if (!K.header.empty()) { out.put(K.header.data(), K.header.size()); if (spl) spl->put(K.header.data(), K.header.size()); if (disc) disc->put(K.header.data(), K.header.size()); wrote_header = true; }
		const std::vector<int64_t> &offs = K.offs;
		const size_t n_blocks = offs.size() - 1;
		if (!n_blocks) continue;
		const std::vector<lrec_t> &L = K.L; const std::vector<ssg_sbl_line_t> &N = K.N; const std::vector<char> &buf = K.buf;
		const size_t nl = (size_t)offs[n_blocks];
		bits.resize(nl); mate.resize(nl);
		if (ssg_sbl_process(st, &o, (long)n_blocks, offs.data(), N.data(), bits.data(), mate.data())) { fprintf(stderr, "[samblaster] %s\n", ssg_last_error()); exit(1); }
		auto emit = [&](out_t &w, const lrec_t &r, int flag, bool patch_flag, const char *suffix, const lrec_t *m, bool add_mc, bool add_mq) {
			const char *s = buf.data() + r.off;
			if (!patch_flag && !suffix) w.put(s, r.len);
			else {
				w.put(s, r.qn_len); if (suffix) w.put(suffix, 2);
				w.putc('\t'); w.puti(flag);
				w.put(s + r.flag_end, r.len - r.flag_end);
			}
			if (m) {
				const char *ms = buf.data() + m->off;
				if (add_mc) { w.put("\tMC:Z:", 6); w.put(ms + m->cig_off, m->cig_len); }
				if (add_mq) { w.put("\tMQ:i:", 6); w.put(ms + m->mq_off, m->mq_len); }
			}
			w.putc('\n');
		};
		auto tags = [&](int64_t i, const lrec_t *&m, bool &add_mc, bool &add_mq) {
			const lrec_t &r = L[i];
			m = (o.add_mate_tags && mate[i] >= 0) ? &L[mate[i]] : 0; add_mc = add_mq = false;
			if (m) { const char *op = r.opt_off ? buf.data() + r.off + r.opt_off : 0, *oe = buf.data() + r.off + r.len; add_mc = !(op && has_tag(op, oe, "MC:Z:")); add_mq = !(op && has_tag(op, oe, "MQ:i:")); }
		};
		for (size_t b = 0; b < n_blocks; ++b) {
			bool paired = false, dup = false; int64_t d1 = -1, d2 = -1;
			for (int64_t i = offs[b]; i < offs[b + 1]; ++i) {
				const int bt = bits[i];
				if (mate[i] >= 0) paired = true;
				if (bt & SSG_SBL_DUP) dup = true;
				if (bt & SSG_SBL_DISC) { if (N[i].flag & 0x40) d1 = i; else d2 = i; }
				const lrec_t *m; bool add_mc, add_mq; tags(i, m, add_mc, add_mq);
				emit(out, L[i], N[i].flag | ((bt & SSG_SBL_DUP) ? 0x400 : 0), (bt & SSG_SBL_DUP) != 0, 0, m, add_mc, add_mq);
			}
			if (paired) { ++n_pairs; if (dup) ++n_dups; }
			if (disc && d1 >= 0 && d2 >= 0) {   /* upstream order: read 1's primary, then read 2's */
				for (int64_t i : { d1, d2 }) { const lrec_t *m; bool add_mc, add_mq; tags(i, m, add_mc, add_mq); emit(*disc, L[i], N[i].flag | (dup ? 0x400 : 0), dup, 0, m, add_mc, add_mq); }
				++n_disc;
			}
			if (spl) for (int64_t i = offs[b]; i < offs[b + 1]; ++i) if (bits[i] & SSG_SBL_SPLIT) {
				const lrec_t *m; bool add_mc, add_mq; tags(i, m, add_mc, add_mq);
				emit(*spl, L[i], N[i].flag | (dup ? 0x400 : 0), true, (N[i].flag & 0x40) ? "_1" : "_2", m, add_mc, add_mq); ++n_spl;
			}
		}
	}
	reader.join();
	if (parse_fail) return 1;
	(void)wrote_header;
	if (in_header) { out.put(pg, strlen(pg)); if (spl) spl->put(pg, strlen(pg)); if (disc) disc->put(pg, strlen(pg)); }
	out.flush();
	if (spl) { spl->flush(); fclose(splf); }
	if (disc) { disc->flush(); fclose(discf); }
	ssg_sbl_state_free(st);
	fprintf(stderr, "[samblaster] pairs=%llu dups=%llu discordant_pairs=%llu splitter_lines=%llu (decisions on %s)\n", n_pairs, n_dups, n_disc, n_spl, ssg_backend());
	return 0;
}
