/*
 * bwa_main.cpp -- the `bwa` executable speedseq.config names (reference bin/speedseq.config:13,
 * invoked at bin/speedseq:389,438,468): `bwa index <ref>` and
 * `bwa mem -t INT [-p] [-I F[,F[,I[,I]]]] -R STR [-C] <ref> <fq1> [fq2]` -> SAM on stdout.
 * Host side of the drop-in boundary: FASTQ parsing, upstream's batch boundaries and SAM printing;
 * all alignment work happens on the MI355X through libssgpu's C ABI (include/ssgpu.h).
 * Mirrors upstream main_mem() (fastmap.c): bseq_read(chunk_size * n_threads) -> mem_process_seqs
 * -> fputs(sam).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>
#include <string>
#include <vector>
#include <algorithm>
#include <thread>
#include <memory>
#include <zlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>
#include <map>
#include <shared_mutex>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include "fastq.h"
#include "fused.h"
#include "ranksplit.h"
#include "rawfeed.h"
#include "bam2sam.h"
#include "../../include/ssgpu.h"

#include <time.h>
static double wall() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void die(const char *what) { fprintf(stderr, "[bwa] %s: %s\n", what, ssg_last_error()); exit(1); }

static std::string unescape(const char *s)
{	/* upstream bwa_set_rg: "\t" -> TAB */
	std::string o;
	for (const char *p = s; *p; ++p) {
		if (*p == '\\' && p[1] == 't') { o += '\t'; ++p; }
		else if (*p == '\\' && p[1] == 'n') { o += '\n'; ++p; }
		else if (*p == '\\' && p[1] == '\\') { o += '\\'; ++p; }
		else o += *p;
	}
	return o;
}

static int main_index(int argc, char **argv)
{	/* upstream bwa_index (bwtindex.c): `bwa index [-p prefix] [-a algo] <in.fasta>`; the FM-index is built on the GPU by libssgpu (ssg_index_build_fasta) */
	const char *prefix = 0; int ai = 1;
	for (; ai < argc && argv[ai][0] == '-' && argv[ai][1]; ++ai) {
		if (!strcmp(argv[ai], "-p") && ai + 1 < argc) prefix = argv[++ai];
		else if (!strcmp(argv[ai], "-a") && ai + 1 < argc) ++ai;          /* construction algorithm: irrelevant here, same bytes */
		else if (!strcmp(argv[ai], "-b") && ai + 1 < argc) ++ai;
		else if (!strcmp(argv[ai], "-6")) ;
		else { fprintf(stderr, "[bwa] index: unsupported option %s\n", argv[ai]); return 1; }
	}
	if (ai >= argc) { fprintf(stderr, "usage: bwa index [-p prefix] <ref.fa>\n"); return 1; }
	ssg_index_t *idx;
	if (ssg_index_build_fasta(argv[ai], &idx)) die("index construction failed");
	if (ssg_index_save(idx, prefix ? prefix : argv[ai])) die("writing the index failed");
	fprintf(stderr, "[bwa] index: %lld bp in %d sequence(s) indexed on %s\n", (long long)ssg_index_l_pac(idx), ssg_index_n_ctg(idx), ssg_backend());
	ssg_index_destroy(idx);
	return 0;
}

static int main_mem(int argc, char **argv)
{
	ssg_mem_opt_t opt; ssg_mem_opt_init(&opt);
	bool interleaved = false, keep_comment = false; std::string rg; char rg_id[256] = "";
	ssg_pestat_t pes0[4], *pes = 0;
	int ai, c, fixed_chunk = 0; char *p;
	struct { bool a, b, T, o_del, e_del, o_ins, e_ins, zdrop, clip5, clip3, unpaired; } set = {};   /* upstream main_mem's opt0: the scoring fields the command line set itself */
	auto second = [&p]() { return *p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1]); };   /* upstream's "INT[,INT]" forms */
	optind = 1;
	/* upstream main_mem's option letters (fastmap.c, 0.7.12).  Not taken: -a -e -F -V -j -x -H (INTEGRATION.md says what each would need) */
	while ((c = getopt(argc, argv, "pMCSPYk:c:v:s:r:t:R:A:B:O:E:U:w:L:d:T:Q:D:m:I:N:W:G:h:y:K:X:")) >= 0) {
		if (c == 'p') interleaved = true;
		else if (c == 'C') keep_comment = true;
		else if (c == 'M') opt.flag |= SSG_F_NO_MULTI;
		else if (c == 'Y') opt.flag |= SSG_F_SOFTCLIP;
		else if (c == 'S') opt.flag |= SSG_F_NO_RESCUE;
		else if (c == 'P') opt.flag |= SSG_F_NOPAIRING;
		else if (c == 'k') opt.min_seed_len = atoi(optarg);
		else if (c == 'w') opt.w = atoi(optarg);
		else if (c == 'A') opt.a = atoi(optarg), set.a = true;
		else if (c == 'B') opt.b = atoi(optarg), set.b = true;
		else if (c == 'T') opt.T = atoi(optarg), set.T = true;
		else if (c == 'U') opt.pen_unpaired = atoi(optarg), set.unpaired = true;
		else if (c == 't') opt.n_threads = atoi(optarg);
		else if (c == 'c') opt.max_occ = atoi(optarg);
		else if (c == 'd') opt.zdrop = atoi(optarg), set.zdrop = true;
		else if (c == 'v') ;   /* verbosity: nothing on stdout depends on it */
		else if (c == 'r') opt.split_factor = (float)atof(optarg);
		else if (c == 'D') opt.drop_ratio = (float)atof(optarg);
		else if (c == 'm') opt.max_matesw = atoi(optarg);
		else if (c == 's') opt.split_width = atoi(optarg);
		else if (c == 'G') opt.max_chain_gap = atoi(optarg);
		else if (c == 'N') opt.max_chain_extend = atoi(optarg);
		else if (c == 'W') opt.min_chain_weight = atoi(optarg);
		else if (c == 'y') opt.max_mem_intv = (uint64_t)atol(optarg);
		else if (c == 'K') fixed_chunk = atoi(optarg);
		else if (c == 'X') opt.mask_level = (float)atof(optarg);
		else if (c == 'h') { opt.max_XA_hits = opt.max_XA_hits_alt = (int)strtol(optarg, &p, 10); if (second()) opt.max_XA_hits_alt = (int)strtol(p + 1, &p, 10); }
		else if (c == 'Q') { opt.mapQ_coef_len = (float)atoi(optarg); opt.mapQ_coef_fac = opt.mapQ_coef_len > 0 ? (int)log(opt.mapQ_coef_len) : 0; }
		else if (c == 'O') { set.o_del = set.o_ins = true; opt.o_del = opt.o_ins = (int)strtol(optarg, &p, 10); if (second()) opt.o_ins = (int)strtol(p + 1, &p, 10); }
		else if (c == 'E') { set.e_del = set.e_ins = true; opt.e_del = opt.e_ins = (int)strtol(optarg, &p, 10); if (second()) opt.e_ins = (int)strtol(p + 1, &p, 10); }
		else if (c == 'L') { set.clip5 = set.clip3 = true; opt.pen_clip5 = opt.pen_clip3 = (int)strtol(optarg, &p, 10); if (second()) opt.pen_clip3 = (int)strtol(p + 1, &p, 10); }
		else if (c == 'R') rg = unescape(optarg);
		else if (c == 'I') { /* upstream main_mem -I: FR orientation only */
			pes = pes0; memset(pes0, 0, sizeof(pes0));
			pes0[0].failed = pes0[2].failed = pes0[3].failed = 1;
			pes0[1].avg = strtod(optarg, &p);
			pes0[1].std = pes0[1].avg * .1;
			if (second()) pes0[1].std = strtod(p + 1, &p);
			pes0[1].high = (int)(pes0[1].avg + 4. * pes0[1].std + .499);
			pes0[1].low  = (int)(pes0[1].avg - 4. * pes0[1].std + .499);
			if (pes0[1].low < 1) pes0[1].low = 1;
			if (second()) pes0[1].high = (int)(strtod(p + 1, &p) + .499);
			if (second()) pes0[1].low  = (int)(strtod(p + 1, &p) + .499);
		} else { fprintf(stderr, "[bwa] mem: option not supported here (of upstream's letters -a -e -F -V -j -x -H are not taken)\n"); return 1; }
	}
	ai = optind;
	if (set.a) {   /* upstream update_a: a changed match score scales the penalties the command line left alone */
		if (!set.b) opt.b *= opt.a;
		if (!set.T) opt.T *= opt.a;
		if (!set.o_del) opt.o_del *= opt.a;
		if (!set.e_del) opt.e_del *= opt.a;
		if (!set.o_ins) opt.o_ins *= opt.a;
		if (!set.e_ins) opt.e_ins *= opt.a;
		if (!set.zdrop) opt.zdrop *= opt.a;
		if (!set.clip5) opt.pen_clip5 *= opt.a;
		if (!set.clip3) opt.pen_clip3 *= opt.a;
		if (!set.unpaired) opt.pen_unpaired *= opt.a;
	}
	for (int i = 0, k = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) opt.mat[k++] = (int8_t)(i == j ? opt.a : -opt.b); opt.mat[k++] = -1; }   /* upstream bwa_fill_scmat */
	if (argc - ai < 2) { fprintf(stderr, "usage: bwa mem [-t INT] [-p] [-I ...] [-R STR] [-C] <ref> <fq1> [fq2]\n"); return 1; }
	if (opt.n_threads < 1) opt.n_threads = 1;
	if (!rg.empty()) {
		if (rg.compare(0, 3, "@RG") != 0) { fprintf(stderr, "[E::bwa_set_rg] the read group line is not started with @RG\n"); return 1; }
		size_t p = rg.find("\tID:");
		if (p == std::string::npos) { fprintf(stderr, "[E::bwa_set_rg] no ID at the read group line\n"); return 1; }
		size_t i = 0; for (p += 4; p < rg.size() && rg[p] != '\t' && rg[p] != '\n' && i < 255; ++p) rg_id[i++] = rg[p];
		rg_id[i] = 0;
	}
	const double t_start = wall();
	ssg_stamp("bwa", "start");
	size_t max_pairs_per_call = 1u << 19;   /* upstream batches are grouped up to this many pairs per device call (one batch alone may exceed it) */
	{ const char *e = getenv("SSG_BWA_CALL_PAIRS"); if (e && atol(e) > 0) max_pairs_per_call = (size_t)atol(e); }
	/* Rank mode (bin/speedseq-ranks): SSG_WORLD pipelines of the reference's script run side by side, one per GPU; this
	 * `bwa mem` reads the whole input, forms upstream's batches as ever -- their composition is the scope of the insert-size model -- and aligns
	 * those whose index is SSG_RANK modulo SSG_WORLD, one batch per device call, pair ordinals counted over the whole input. */
	int world = 1, rank = 0;
	{ const char *w = getenv("SSG_WORLD"), *r = getenv("SSG_RANK"); if (w && atoi(w) > 1) { world = atoi(w); rank = r ? atoi(r) : -1; if (rank < 0 || rank >= world) { fprintf(stderr, "[bwa] SSG_RANK must be 0 .. SSG_WORLD - 1\n"); return 1; } } }
	if (world > 1) max_pairs_per_call = 1;
	if (getenv("SSG_BWA_PROF")) { ssg_prof_reset(); ssg_prof_enable(1); }
	/* fused mode (fused.h): BAM records in frames instead of SAM text when speedseq.config exported SSG_FUSED=1; never with -C */
	bool fused = fu_enabled() && !keep_comment;
	gzFile fp1 = gzopen(argv[ai + 1], "r"), fp2 = 0;
	if (!fp1) { fprintf(stderr, "[bwa] fail to open %s\n", argv[ai + 1]); return 1; }
	if (argc - ai >= 3) { fp2 = gzopen(argv[ai + 2], "r"); if (!fp2) { fprintf(stderr, "[bwa] fail to open %s\n", argv[ai + 2]); return 1; } }
	if (interleaved && fp2) { fprintf(stderr, "[W::main_mem] when '-p' is in use, the second query file is ignored.\n"); gzclose(fp2); fp2 = 0; }
	const bool se = !interleaved && !fp2;   /* upstream main_mem: MEM_F_PE is set by -p or by a second file; without it every read is aligned on its own */
	if (se && world > 1) { fprintf(stderr, "[bwa] rank mode is for paired-end input\n"); return 1; }
	if (world > 1 && !fused) { fprintf(stderr, "[bwa] rank mode needs the fused hand-off: `export SSG_FUSED=1` in speedseq.config (and no -C)\n"); return 1; }
	if (se) fused = false;                  /* samblaster has nothing to do with unpaired reads: SAM text */
	{ const char *e = getenv("SSG_BWA_CHUNK_BASES"); if (e && atoi(e) > 0) opt.chunk_size = atoi(e); }   /* tests: upstream's 10 M bases per thread make a batch of 33 k pairs */
	const int64_t chunk = fixed_chunk > 0 ? fixed_chunk : (int64_t)opt.chunk_size * opt.n_threads;   /* -K: batches that do not depend on -t */

	/* One worker thread per visible device (SURVEY 8e coupling 1: whole upstream batches go to the GPUs, no collective): each loads its
	 * own replica of the index and takes the next assembled batch when it is free; results are re-serialised in input order. */
	int n_dev = ssg_device_count();
	{ const char *e = getenv("SSG_BWA_DEVICES"); if (e && atoi(e) > 0) n_dev = std::min(n_dev, atoi(e)); }
	if (n_dev < 1) { fprintf(stderr, "[bwa] no MI355X visible: %s has no CPU path\n", ssg_backend()); return 1; }
	if (n_dev > 16) n_dev = 16;
	/* the readers (inflate + parse) start now: the first batches are parsed while the index travels to the device(s); several devices
	 * take proportionally more parse threads on plain files (one thread parses about what one MI355X aligns) */
	const int parse_hint = n_dev > 1 ? std::min(48, (fp2 ? 3 : 5) * n_dev) : 0;
	/* rank mode on plain regular files (ranksplit.h): rank 0 scans the input for upstream's batches and publishes their byte ranges; every rank parses
	 * the ranges of its batches only; compressed input is inflated and scanned by rank 0 alone and handed on batch by batch.  SSG_RANKS_SPLIT=0: every
	 * rank reads and parses everything and keeps its batches. */
	std::atomic<int> fail(0);
	const bool split = world > 1 && !(getenv("SSG_RANKS_SPLIT") && !strcmp(getenv("SSG_RANKS_SPLIT"), "0"));
	const bool served = split && !(rs_plain_regular(argv[ai + 1]) && (!fp2 || rs_plain_regular(argv[ai + 2])));   /* compressed input: rank 0 inflates, scans and hands the batches on as files */
	std::shared_ptr<rs_table_t> rs_tab(split ? new rs_table_t(rk_dir()) : 0);
	std::thread t_scan;
	if (split && !rk_check("bwa")) return 1;
	if (split && rank == 0) { const std::string f1 = argv[ai + 1], f2 = fp2 ? argv[ai + 2] : ""; const std::string rdv = rk_dir(); t_scan = std::thread([f1, f2, rdv, chunk, served, world, &fail]() { if (!rs_scan_and_publish(rdv, f1.c_str(), f2.empty() ? 0 : f2.c_str(), chunk, served, world)) fail = 1; }); }
	struct scan_join_t { std::thread &t; ~scan_join_t() { if (t.joinable()) t.join(); } } scan_join = { t_scan };
	/* The device-text path (rawfeed.h, csrc/k_bam.h; SURVEY 2.1 K1 + K11): with the fused hand-off the reads travel to the MI355X as the FASTQ text they
	 * are -- the host only finds the records by their newlines and forms upstream's batches from the sequence lengths -- and come back as the BAM
	 * records of their alignments; no parser, no formatter on the host.  Input that is not plain four-line records goes through the parser from the
	 * batch where the scanner meets it.  SSG_BWA_DEVTEXT=0: parser and host formatter as before (the three BAMs are the same either way: tests/test_fused.py). */
	const bool devtext = fused && world == 1 && !(getenv("SSG_BWA_DEVTEXT") && !strcmp(getenv("SSG_BWA_DEVTEXT"), "0"));
	std::unique_ptr<raw_feed_t> rawf(devtext ? new raw_feed_t(argv[ai + 1], fp1, fp2 ? argv[ai + 2] : 0, fp2, chunk, max_pairs_per_call) : 0);
	std::unique_ptr<fq_feed_t> feed1(devtext ? 0 : split ? new fq_feed_t(rs_provider(rs_tab, argv[ai + 1], false, rank, world, &fail, served), keep_comment, 16384, parse_hint)
	                                         : new fq_feed_t(fp1, keep_comment, 16384, argv[ai + 1], parse_hint));
	std::unique_ptr<fq_feed_t> feed2(!fp2 || devtext ? 0 : split ? new fq_feed_t(rs_provider(rs_tab, argv[ai + 2], true, rank, world, &fail, served), keep_comment, 16384, parse_hint)
	                                                  : new fq_feed_t(fp2, keep_comment, 16384, argv[ai + 2], parse_hint));
	/* Overlapped stages, one batch each: (1) assemble upstream's batches from the reader threads' blocks, (2) align on an MI355X,
	 * (3) format (SAM text, or BAM records in fused mode; threads inside libssgpu) and (4) write.  Several upstream batches
	 * (bseq_read's chunk_size * n_threads bases, even read count: the scope of the insert-size model) travel to the GPU in one call. */
	struct batch_t {   /* names, comments and qualities stay where the reader put them (the batch holds on to those blocks); the bases are gathered
	                    * for the device when the batch is complete, by several threads (the blocks were written by another core) */
		std::vector<std::shared_ptr<fq_block_t> > hold;
		std::vector<const char*> names, quals, comments;      /* 0 = absent */
		std::vector<const uint8_t*> src;
		std::unique_ptr<uint8_t[]> seq; std::vector<int64_t> off;
		std::vector<int32_t> pair_batch; int n_batches; int64_t id0, seqno;
		ssg_pe_result_t *res; int dev;
		std::unique_ptr<raw_call_t> raw; ssg_pe_bam_t *bres;   /* the device-text path: the batch is FASTQ text, its result BAM record bytes */
		batch_t() : n_batches(0), id0(0), seqno(0), res(0), dev(0), bres(0) { off.push_back(0); }
		int n() const { return raw ? 2 * raw->n_pairs() : (int)names.size(); }
		void add(const std::shared_ptr<fq_block_t> &h, int i)
		{
			const fq_block_t &b = *h;
			if (hold.empty() || hold.back().get() != h.get()) {
				bool seen = false;                                  /* two input files alternate between two blocks */
				for (size_t k = hold.size(); k-- > 0 && k + 4 > hold.size(); ) if (hold[k].get() == h.get()) { seen = true; break; }
				if (!seen) hold.push_back(h);
			}
			names.push_back(b.txt.data() + b.name_o[i]);
			comments.push_back(b.com_o[i] != UINT32_MAX ? b.txt.data() + b.com_o[i] : 0);
			quals.push_back(b.has_q[i] ? b.qual.data() + b.qual_o[i] : 0);
			src.push_back(b.seq.data() + b.seq_o[i]);
			off.push_back(off.back() + (int64_t)(b.seq_o[i + 1] - b.seq_o[i]));
		}
		void drop_last() { names.pop_back(); comments.pop_back(); quals.pop_back(); src.pop_back(); off.pop_back(); }
		void gather(int n_threads)
		{
			const size_t nr = src.size();
			seq.reset(new uint8_t[(size_t)off[nr] + 1]);              /* not value-initialised: first touched by the copying threads */
			const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, nr / 65536 + 1));
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([this, t, T, nr]() {
				for (size_t r = nr * (size_t)t / (size_t)T, e = nr * (size_t)(t + 1) / (size_t)T; r < e; ++r) memcpy(seq.get() + off[r], src[r], (size_t)(off[r + 1] - off[r]));
			});
			for (std::thread &x : th) x.join();
		}
	};
	/* SSG_BWA_INFLIGHT = k > 1: k worker threads per device, each on a lane of its own (ssg_set_lane: own stream, own arena), so that the
	 * upload and download of one call run under the kernels of another; default 2 since round 4 (MI355X, 8 M pairs: reads -> records 4.72 -> 4.11 s,
	 * the three BAMs unchanged; profiles/r04h); 1 = one call at a time on the default stream */
	int inflight = 2; { const char *e = getenv("SSG_BWA_INFLIGHT"); if (e && atoi(e) > 0) inflight = std::min(3, atoi(e)); }
	const int n_work = n_dev * inflight;
	chan_t<std::unique_ptr<batch_t> > to_gpu((size_t)n_work);
	/* aligned batches wait here for their turn: the formatter takes them in input order whichever device finished first */
	struct reorder_t {
		std::mutex mu; std::condition_variable cv; std::map<int64_t, std::unique_ptr<batch_t> > ready; int64_t next; int open_workers; size_t cap;
		const std::atomic<int> *failed;
		reorder_t(int w, size_t c, const std::atomic<int> *f) : next(0), open_workers(w), cap(c), failed(f) {}
		/* once the run has failed the batch the buffer is waiting for may never come (its worker gave up): a worker must not wait for room then,
		 * or it never reports itself done and the formatters wait for it for ever.  `fail' is raised in many places without this lock, hence the timed wait. */
		void put(std::unique_ptr<batch_t> B)
		{
			std::unique_lock<std::mutex> l(mu); const int64_t s = B->seqno;
			while (!(s == next || ready.size() < cap || failed->load())) cv.wait_for(l, std::chrono::milliseconds(50));
			if (failed->load() && !(s == next || ready.size() < cap)) { if (B->res) ssg_pe_result_free(B->res); if (B->bres) ssg_pe_bam_free(B->bres); return; }
			ready[s] = std::move(B); cv.notify_all();
		}
		void worker_done() { std::lock_guard<std::mutex> l(mu); --open_workers; cv.notify_all(); }
		bool take(std::unique_ptr<batch_t> &B)
		{
			std::unique_lock<std::mutex> l(mu);
			cv.wait(l, [&] { return (!ready.empty() && ready.begin()->first == next) || open_workers == 0; });
			if (ready.empty() || ready.begin()->first != next) {
				if (ready.empty()) return false;
				next = ready.begin()->first;             /* a batch was lost to an error upstream: drain what is left */
			}
			B = std::move(ready.begin()->second); ready.erase(ready.begin()); ++next; cv.notify_all(); return true;
		}
	} to_fmt(n_work, (size_t)n_work + 1, &fail);
	double tm_asm = 0; std::vector<double> tm_gpu((size_t)n_dev, 0.0); std::vector<long> calls((size_t)n_dev, 0);
	/* the parser's side of the assembler: upstream's batches from the reader threads' blocks, from pair ordinal id0 / device call seqno on */
	auto parsed_asm = [&](fq_feed_t &feed1, fq_feed_t *feed2, int64_t id0, int64_t seqno) {
		fq_cursor_t c1(feed1); std::unique_ptr<fq_cursor_t> c2(feed2 ? new fq_cursor_t(*feed2) : 0);
		int64_t bidx = 0; bool eof = false;
		while (!eof && !fail) {
			const double t0 = wall();
			std::unique_ptr<batch_t> B(new batch_t()); B->id0 = id0; B->seqno = seqno;
			while (se && !eof && (size_t)B->n() < 2 * max_pairs_per_call) {   /* single-end: a read's result depends on its ordinal only, batches need not be kept apart */
				const fq_block_t *ba; int ia = 0;
				(void)ba;
				const int rc = c1.next(&ba, &ia);
				if (rc == -1) { eof = true; break; }
				if (rc < 0) { fprintf(stderr, "[bwa] truncated or malformed FASTQ\n"); fail = 1; eof = true; break; }
				B->add(c1.cur, ia);
			}
			while (!se && !eof && (size_t)B->n() / 2 < max_pairs_per_call) {   /* upstream bseq_read: one batch */
				int64_t size = 0; const int n0 = B->n();
				for (;;) {
					const fq_block_t *ba, *bb; int ia = 0, ib = 0;
					(void)ba; (void)bb;
					int rc = c1.next(&ba, &ia);
					if (rc == -1) { eof = true; break; }
					if (rc < 0) { fprintf(stderr, "[bwa] truncated or malformed FASTQ\n"); fail = 1; eof = true; break; }
					B->add(c1.cur, ia);
					rc = (c2 ? *c2 : c1).next(&bb, &ib);
					if (rc == -1) {   /* upstream bseq_read / main_mem: the complete pairs read so far are aligned and printed, the odd read is dropped */
						fprintf(stderr, c2 ? "[W::bseq_read] the 2nd file has fewer sequences.\n" : "[W::main_mem] odd number of reads in the PE mode; last read dropped\n");
						B->drop_last(); eof = true; break;
					}
					if (rc < 0) { fprintf(stderr, "[bwa] truncated or malformed FASTQ (paired reads expected)\n"); fail = 1; eof = true; break; }
					B->add((c2 ? *c2 : c1).cur, ib);
					const int n = B->n();
					if (strcmp(B->names[n - 2], B->names[n - 1]) != 0) {
						fprintf(stderr, "[mem_sam_pe] paired reads have different names: \"%s\", \"%s\"\n", B->names[n - 2], B->names[n - 1]); fail = 1; eof = true; break; }
					size += (B->off[n - 1] - B->off[n - 2]) + (B->off[n] - B->off[n - 1]);
					if (size >= chunk) break;
				}
				if (fail) break;
				if (B->n() > n0) { for (int p = n0 / 2; p < B->n() / 2; ++p) B->pair_batch.push_back(B->n_batches); ++B->n_batches; }
			}
			if (fail || B->n() == 0) break;
			if (split) {   /* every batch of this stream is this rank's: its place in the input comes from the scanner's table */
				rs_entry_t e; uint64_t first_pair = 0;
				if (rs_tab->get((uint64_t)rank + (uint64_t)bidx * (uint64_t)world, &e, &first_pair) != 1 || e.pairs != (uint64_t)(B->n() / 2)) {
					fprintf(stderr, "[bwa] rank mode: batch %lld of this rank is not what the scan of the input found (%d pairs parsed)\n", (long long)bidx, B->n() / 2); fail = 1; break; }
				++bidx; B->id0 = (int64_t)first_pair;
			} else if (world > 1 && (bidx++ % world) != rank) { id0 += B->n() / 2; continue; }   /* another rank's batch: only its pairs are counted */
			B->gather(std::min(8, std::max(1, opt.n_threads)));
			id0 += se ? B->n() : B->n() / 2; ++seqno;   /* pairs: the pair ordinal; single-end: upstream's n_processed */
			tm_asm += wall() - t0;
			to_gpu.push(std::move(B));
		}
	};
	std::unique_ptr<fq_feed_t> fb1, fb2;   /* the parser taking over from the device-text path (an input that is not plain four-line records) */
	/* The assembler waits for the index: page-locking its text buffers (and the result blocks of ssg_pe_reserve) while the index files stream into HBM made that load
	 * three times as long -- 0.89 s against 0.34 s alone, the driver serialises page-locked allocations and the loader's staging blocks queued behind gigabytes of
	 * them -- and nothing the assembler prepares can be used before the index is there (profiles/r06y_literal_load_contention.json).  SSG_BWA_LATE_PARSE=0: as before. */
	std::atomic<bool> go_parse(getenv("SSG_BWA_LATE_PARSE") && atoi(getenv("SSG_BWA_LATE_PARSE")) == 0);
	std::thread t_asm([&]() {
		while (!go_parse.load() && !fail) std::this_thread::sleep_for(std::chrono::microseconds(200));
		if (rawf) {
			int64_t id0 = 0, seqno = 0;
			while (!fail) {
				const double t0 = wall();
				std::unique_ptr<batch_t> B(new batch_t()); B->raw.reset(new raw_call_t());
				const int rc = rawf->next_call(B->raw.get());
				if (rc < 0) { fprintf(stderr, "[bwa] %s\n", rawf->msg.c_str()); fail = 1; break; }
				tm_asm += wall() - t0;
				if (rc == 0) break;
				B->id0 = id0; B->seqno = seqno++; B->n_batches = B->raw->n_batches; id0 += B->raw->n_pairs();
				to_gpu.push(std::move(B));
			}
			if (!fail && rawf->fell_back) {
				fprintf(stderr, "[bwa] %s after %llu plain pairs: the parser takes the rest of the input\n", rawf->why.c_str(), (unsigned long long)rawf->pairs_done);
				raw_feed_t *rf = rawf.get();
				auto mk = [rf, keep_comment](int i) -> std::function<fq_reader_t*()> {
					return [rf, i, keep_comment]() -> fq_reader_t* {
						raw_src_t &S = i ? *rf->B : *rf->A;
						if (S.p) { struct stat sb; if (S.fd < 0 || fstat(S.fd, &sb) != 0) return 0; return new fq_reader_t(S.fd, rf->resume_off[i], (size_t)sb.st_size, keep_comment); }
						return new fq_reader_t(rf->resume_mem[i].data(), rf->resume_mem[i].size(), S.ks.get(), keep_comment);
					};
				};
				fb1.reset(new fq_feed_t(mk(0), 16384));
				if (rf->B) fb2.reset(new fq_feed_t(mk(1), 16384));
				parsed_asm(*fb1, fb2.get(), id0, seqno);
			}
		} else parsed_asm(*feed1, feed2.get(), 0, 0);
		to_gpu.close();
	});
	/* the index travels to the device(s) while the assembler is already at work: by the time it is there the first device calls are waiting (until round 6 the
	 * assembler started afterwards and the first call, alone on the device, began a scan of half a million pairs later) */
	std::vector<ssg_index_t*> idxs((size_t)n_dev, (ssg_index_t*)0);
	/* The denser suffix-array copy (every 4th row instead of the file's every 32nd) takes locating the seeds of a million pairs from 67 ms to 8 ms and costs one LF
	 * walk over the text: 0.21 s on the MI355X at 3.1 Gbp since the walk's lanes refill in batches (1.05 s before, when it was put off until an input had proved
	 * long) -- paid back after 4 M pairs.  So it is made at load time, unless the input is a plain regular file that is known to be shorter than that.
	 * SSG_BWA_DENSIFY_AFTER=n: after n pairs instead (0 = at load time).  Results do not depend on it. */
	long densify_after = 0; { const char *e = getenv("SSG_BWA_DENSIFY_AFTER"); if (e) densify_after = atol(e); }
	int sa_first = 32; { const char *e = getenv("SSG_BWA_SA_FIRST"); if (e && atoi(e) > 0) sa_first = atoi(e); }   /* (experiment: a first step to a sparser copy; costs the same walk) */
	if (!getenv("SSG_BWA_DENSIFY_AFTER")) {
		uint64_t bytes = 0; bool known = true;
		for (int k = ai + 1; k < argc && k < ai + 3; ++k) {
			struct stat sb; const size_t l = strlen(argv[k]);
			if (stat(argv[k], &sb) != 0 || !S_ISREG(sb.st_mode) || (l > 3 && !strcmp(argv[k] + l - 3, ".gz"))) { known = false; break; }
			bytes += (uint64_t)sb.st_size;
		}
		if (known && bytes / 640 < 4000000) densify_after = 4000000;   /* (about 640 bytes of FASTQ a pair at 2x150: never reached) */
	}
	const int warm = getenv("SSG_BWA_WARM") ? atoi(getenv("SSG_BWA_WARM")) : 0;   /* page-locked result blocks ahead of the first calls: 1 = while the index loads (until round 6), 2 = right after; 0 = the calls make them as they go */
	std::thread t_warm([max_pairs_per_call, warm]() { if (warm == 1) (void)ssg_pe_reserve((int)std::min<size_t>(max_pairs_per_call, (size_t)1 << 22), 2); });   /* page-locked result blocks, while the index loads */
	{
		std::vector<std::thread> ld;
		for (int g = 0; g < n_dev; ++g) ld.emplace_back([&, g]() {
			if (ssg_set_device(g) || ssg_index_load2(argv[ai], densify_after > 0, &idxs[(size_t)g])) { fprintf(stderr, "[bwa] fail to load the index on device %d: %s\n", g, ssg_last_error()); fail = 1; }
			else if (densify_after > 0 && sa_first < 32 && ssg_index_densify_to(idxs[(size_t)g], sa_first)) { fprintf(stderr, "[bwa] %s\n", ssg_last_error()); fail = 1; } });
		for (std::thread &x : ld) x.join();
	}
	t_warm.join();
	go_parse = true;
	if (warm == 2) t_warm = std::thread([max_pairs_per_call]() { (void)ssg_pe_reserve((int)std::min<size_t>(max_pairs_per_call, (size_t)1 << 22), 2); });
	ssg_index_t *idx = idxs[0];
	const double t_loaded = wall();
	ssg_stamp("bwa", "index_loaded");
	/* header: upstream bwa_print_sam_hdr + @PG */
	std::string hdr;
	if (!fail) for (int i = 0; i < ssg_index_n_ctg(idx); ++i) { char b[64]; snprintf(b, sizeof(b), "\tLN:%d\n", ssg_index_len(idx, i)); hdr += "@SQ\tSN:"; hdr += ssg_index_name(idx, i); hdr += b; }
	if (!rg.empty()) { hdr += rg; hdr += '\n'; }
	{ hdr += "@PG\tID:bwa\tPN:bwa\tVN:0.7.12-ssgpu\tCL:bwa"; for (int i = 0; i < argc; ++i) { hdr += ' '; hdr += argv[i]; } hdr += '\n'; }
#ifdef F_SETPIPE_SZ
	(void)fcntl(1, F_SETPIPE_SZ, 1 << 20);   /* fewer wake-ups on the pipe to samblaster */
#endif
	if (fused) fu_seg_sweep();
	/* (the assembler is running: a failure from here on is flagged and runs through the common way out, which stops and joins every stage) */
	if (fail) {}
	else if (fused) { if (!fu_write_full(1, FU_MAGIC, 8) || !fu_write_frame(1, FU_HEADER, hdr.data(), hdr.size())) { perror("[bwa] write"); fail = 1; } }
	else if (!fu_write_full(1, hdr.data(), hdr.size())) { perror("[bwa] write"); fail = 1; }

	std::vector<std::thread> t_gpu;
	/* per device: calls hold the index shared, the one-off densification holds it alone */
	struct dev_state_t { std::shared_mutex mu; std::atomic<long> pairs{0}; std::atomic<bool> dense{false}; std::mutex tm; };
	std::vector<std::unique_ptr<dev_state_t> > ds; for (int g = 0; g < n_dev; ++g) { ds.emplace_back(new dev_state_t()); ds.back()->dense = densify_after <= 0; }
	for (int w = 0; w < n_work; ++w) t_gpu.emplace_back([&, w]() {
		const int g = w / inflight, lane = inflight > 1 ? 1 + w % inflight : 0;
		dev_state_t &D = *ds[(size_t)g];
		std::unique_ptr<batch_t> B;
		if (ssg_set_device(g) || ssg_set_lane(lane)) { fprintf(stderr, "[bwa] %s\n", ssg_last_error()); fail = 1; }
		while (to_gpu.pop(B)) {
			const double t0 = wall();
			B->dev = g;
			if (!fail && !D.dense.load() && D.pairs.load() >= densify_after) {
				std::unique_lock<std::shared_mutex> l(D.mu);
				if (!D.dense.load()) {
					if (ssg_index_densify(idxs[(size_t)g])) { fprintf(stderr, "[bwa] %s\n", ssg_last_error()); fail = 1; }
					else fprintf(stderr, "[bwa] device %d: denser suffix-array copy made after %ld pairs (%.2f s)\n", g, D.pairs.load(), wall() - t0);
					D.dense = true;
				}
			}
			D.pairs += B->n() / 2;
			if (B->raw) {	/* FASTQ text in, BAM record bytes out (csrc/ssg_bam.cpp) */
				std::shared_lock<std::shared_mutex> l(D.mu);
				raw_call_t &R = *B->raw;
				const uint8_t *parts[2] = { R.txt[0].p, R.txt[1].p }; const int64_t nb[2] = { (int64_t)R.txt[0].n, (int64_t)R.txt[1].n };
				if (!fail && ssg_mem_process_fastq_bam(idxs[(size_t)g], &opt, R.n_pairs(), parts, nb, R.txt[1].n ? 2 : 1, R.rec_off.data(), R.pair_batch.data(), R.n_batches, B->id0, pes, rg_id, &B->bres)) {
					const char *m = ssg_last_error();
					if (!strncmp(m, "[mem_sam_pe]", 12)) fprintf(stderr, "%s\n", m); else fprintf(stderr, "[bwa] alignment failed on device %d: %s\n", g, m);
					fail = 1; }
				R.txt[0].release(); R.txt[1].release();   /* the text has been read: its page-locked blocks go back to the pool for the next call */
			} else
			{	std::shared_lock<std::shared_mutex> l(D.mu);
				if (!fail && (se ? ssg_mem_process_reads(idxs[(size_t)g], &opt, B->n(), B->seq.get(), B->off.data(), B->id0, &B->res)
				                 : ssg_mem_process_pairs(idxs[(size_t)g], &opt, B->n() / 2, B->seq.get(), B->off.data(), B->pair_batch.data(), B->n_batches, B->id0, pes, &B->res))) {
					fprintf(stderr, "[bwa] alignment failed on device %d: %s\n", g, ssg_last_error()); fail = 1; }
			}
			{ std::lock_guard<std::mutex> l(D.tm); tm_gpu[(size_t)g] += wall() - t0; ++calls[(size_t)g]; }
			if (!fail) to_fmt.put(std::move(B));
		}
		to_fmt.worker_done();
	});
	struct text_t { char *p; size_t len; uint32_t frame; fu_buf_t *seg; std::string *seg_path; };   /* frame: 0 = raw bytes, otherwise the fused frame type to wrap them in; seg: the payload sits in a mapped segment (fused.h REF) */
	chan_t<text_t> to_write(4);
	std::thread t_write([&]() {   /* stdout is a pipe in the reference's pipeline: its reader sets the pace, so writing gets its own thread */
		text_t t;
		while (to_write.pop(t)) {
			if (t.seg) {
				if (fail) { t.seg->reset(); unlink(t.seg_path->c_str()); }
				else if (!fu_seg_send(1, t.frame, *t.seg, *t.seg_path)) { perror("[bwa] write"); fail = 1; }
				delete t.seg; delete t.seg_path;
				continue;
			}
			if (!fail && !(t.frame ? fu_write_frame(1, t.frame, t.p, t.len) : fu_write_full(1, t.p, t.len))) { perror("[bwa] write"); fail = 1; }
			ssg_free(t.p);
		}
	});
	double tm_fmt = 0;
	/* formatters: batches are taken in input order, formatted (threads inside libssgpu), and handed to the writer in input order again;
	 * one formatter keeps up with two devices, so there are more of them when there are more devices (SSG_BWA_FORMATTERS) */
	int n_fmt = n_dev > 2 ? std::min(4, (n_dev + 1) / 2) : 1; { const char *e = getenv("SSG_BWA_FORMATTERS"); if (e && atoi(e) > 0) n_fmt = std::min(8, atoi(e)); }
	struct ordered_t { std::mutex mu; std::map<int64_t, text_t> pend; int64_t next; ordered_t() : next(0) {} } ord;
	auto emit = [&](int64_t seq, const text_t &t) {
		std::lock_guard<std::mutex> l(ord.mu);
		ord.pend[seq] = t;
		while (!ord.pend.empty() && ord.pend.begin()->first == ord.next) { to_write.push(ord.pend.begin()->second); ord.pend.erase(ord.pend.begin()); ++ord.next; }
	};
	auto fmt_loop = [&]() {
		std::unique_ptr<batch_t> B;
		std::vector<int32_t> cand; std::vector<int64_t> sam_off; double busy = 0;
		while (to_fmt.take(B)) {
			if (fail) { if (B->res) ssg_pe_result_free(B->res); if (B->bres) ssg_pe_bam_free(B->bres); continue; }
			const double t0 = wall();
			const int n = B->n();
			text_t t; t.p = 0; t.len = 0; t.frame = 0; t.seg = 0; t.seg_path = 0;
			sam_off.resize((size_t)n + 1);
			if (B->raw) {
				/* the records came from the device as BAM bytes; the SAM text of the pairs samblaster may copy to a side stream is printed from their records
				 * (bam2sam.h: sam_format1's rules -- the line a record came from, since the record is what sam_parse1 makes of that line) */
				const ssg_pe_bam_t *R = B->bres;
				const uint8_t *bam = ssg_pe_bam_data(R); const ssg_bam_cand_t *cd = ssg_pe_bam_cands(R); const size_t nc = (size_t)ssg_pe_bam_n_cand(R);
				std::string ctext; std::vector<uint64_t> c_off(nc + 1, 0); bool ok = true;
				auto cname = [&](int i) { return ssg_index_name(idx, i); };
				ctext.reserve(nc * 1200);
				for (size_t k = 0; k < nc && ok; ++k) {
					c_off[k] = ctext.size();
					const uint8_t *q = bam + cd[k].byte_off;
					for (int64_t j = 0; j < cd[k].n_rec && ok; ++j) { uint32_t bs; memcpy(&bs, q, 4); ok = bam_record_to_sam(q, cname, ctext); q += 4 + (size_t)bs; }
				}
				if (!ok) { fprintf(stderr, "[bwa] a record the device made cannot be printed as SAM\n"); fail = 1; ssg_pe_bam_free(B->bres); continue; }
				fu_batch_t bh; bh.n_rec = (uint64_t)ssg_pe_bam_n_rec(R); bh.bam_bytes = (uint64_t)ssg_pe_bam_bytes(R); bh.n_cand = nc; bh.text_bytes = ctext.size();
				t.len = sizeof(bh) + nc * sizeof(fu_cand_t) + (size_t)bh.text_bytes + (size_t)bh.bam_bytes; t.frame = FU_BATCH;
				{	std::unique_ptr<fu_buf_t> sg(new fu_buf_t()); std::unique_ptr<std::string> sp(new std::string());
					if (fu_seg_create(t.len, *sg, *sp)) { t.seg = sg.release(); t.seg_path = sp.release(); t.p = (char*)t.seg->p; }
					else t.p = (char*)malloc(t.len ? t.len : 1); }
				if (!t.p) { fprintf(stderr, "[bwa] out of memory\n"); fail = 1; ssg_pe_bam_free(B->bres); continue; }
				char *w = t.p; memcpy(w, &bh, sizeof(bh)); w += sizeof(bh);
				for (size_t k = 0; k < nc; ++k) { fu_cand_t c; c.first_rec = (uint64_t)cd[k].first_rec; c.n_rec = (uint64_t)cd[k].n_rec; c.text_off = c_off[k]; memcpy(w, &c, sizeof(c)); w += sizeof(c); }
				if (bh.text_bytes) memcpy(w, ctext.data(), (size_t)bh.text_bytes);
				w += bh.text_bytes;
				{	const size_t nb = (size_t)bh.bam_bytes; const int T = (int)std::max<size_t>(1, std::min<size_t>(8, nb >> 24));
					std::vector<std::thread> th;
					for (int k = 0; k < T; ++k) th.emplace_back([=]() { const size_t a = nb * (size_t)k / (size_t)T, e = nb * (size_t)(k + 1) / (size_t)T; memcpy(w + a, bam + a, e - a); });
					for (std::thread &x : th) x.join();
				}
				busy += wall() - t0;
				const ssg_pestat_t *pp = ssg_pe_bam_pes(R);
				fprintf(stderr, "[bwa] processed %d reads in %d upstream batch(es) on %s device %d; FR insert (first batch): failed=%d low=%d high=%d avg=%.2f std=%.2f\n",
				        n, B->n_batches, ssg_backend(), B->dev, pp[1].failed, pp[1].low, pp[1].high, pp[1].avg, pp[1].std);
				ssg_pe_bam_free(B->bres);
				emit(B->seqno, t);
				continue;
			}
			if (!fused) {
				char *sam;
				if (se ? ssg_sam_format_se(idx, &opt, B->res, n, B->names.data(), B->seq.get(), B->off.data(), B->quals.data(), B->comments.data(), rg_id, &sam, sam_off.data())
				       : ssg_sam_format(idx, &opt, B->res, n / 2, B->names.data(), B->seq.get(), B->off.data(), B->quals.data(), B->comments.data(), rg_id, &sam, sam_off.data())) {
					fprintf(stderr, "[bwa] SAM formatting failed: %s\n", ssg_last_error()); fail = 1; ssg_pe_result_free(B->res); continue; }
				t.p = sam; t.len = (size_t)sam_off[(size_t)n];
			} else {
				/* BAM records of every read + the SAM text of the pairs samblaster may copy to a side stream: a read with supplementary
				 * lines (splitter test) or both ends mapped without the proper-pair flag (discordant test) -- a superset under any options */
				const int64_t *req_off = ssg_pe_req_off(B->res); const ssg_alnreq_t *req = ssg_pe_req(B->res); const ssg_aln_t *alns = ssg_pe_alns(B->res);
				std::vector<int32_t> nmain((size_t)n); cand.clear();
				for (int r = 0; r < n; ++r) { int c = 0; for (int64_t g = req_off[r]; g < req_off[r + 1]; ++g) c += req[g].kind == SSG_REQ_MAIN; nmain[(size_t)r] = c; }
				for (int p = 0; p < n / 2; ++p) {
					const ssg_aln_t &a1 = alns[req_off[2 * p]], &a2 = alns[req_off[2 * p + 1]];
					if (nmain[2 * (size_t)p] > 1 || nmain[2 * (size_t)p + 1] > 1 || (a1.rid >= 0 && a2.rid >= 0 && !(a1.flag & 0x2))) cand.push_back(p);
				}
				uint8_t *bam; char *ctext = 0; std::vector<int64_t> bam_off((size_t)n + 1), c_off(2 * cand.size() + 1, 0);
				if (ssg_bam_format(idx, &opt, B->res, n / 2, B->names.data(), B->seq.get(), B->off.data(), B->quals.data(), rg_id, &bam, bam_off.data())
				    || (!cand.empty() && ssg_sam_format_sel(idx, &opt, B->res, cand.data(), (int)cand.size(), B->names.data(), B->seq.get(), B->off.data(), B->quals.data(), 0, rg_id, &ctext, c_off.data()))) {
					fprintf(stderr, "[bwa] record formatting failed: %s\n", ssg_last_error()); fail = 1; ssg_pe_result_free(B->res); continue; }
				std::vector<int64_t> rec0((size_t)n / 2 + 1, 0);      /* ordinal of each pair's first record within the batch */
				for (int p = 0; p < n / 2; ++p) rec0[(size_t)p + 1] = rec0[(size_t)p] + nmain[2 * (size_t)p] + nmain[2 * (size_t)p + 1];
				fu_batch_t bh; bh.n_rec = (uint64_t)rec0[(size_t)n / 2]; bh.bam_bytes = (uint64_t)bam_off[(size_t)n]; bh.n_cand = cand.size(); bh.text_bytes = (uint64_t)c_off[2 * cand.size()];
				t.len = sizeof(bh) + cand.size() * sizeof(fu_cand_t) + (size_t)bh.text_bytes + (size_t)bh.bam_bytes; t.frame = FU_BATCH;
				{	std::unique_ptr<fu_buf_t> sg(new fu_buf_t()); std::unique_ptr<std::string> sp(new std::string());
					if (fu_seg_create(t.len, *sg, *sp)) { t.seg = sg.release(); t.seg_path = sp.release(); t.p = (char*)t.seg->p; }
					else t.p = (char*)malloc(t.len ? t.len : 1); }
				if (!t.p) { fprintf(stderr, "[bwa] out of memory\n"); fail = 1; ssg_pe_result_free(B->res); continue; }
				char *w = t.p; memcpy(w, &bh, sizeof(bh)); w += sizeof(bh);
				for (size_t k = 0; k < cand.size(); ++k) { fu_cand_t c; c.first_rec = (uint64_t)rec0[(size_t)cand[k]]; c.n_rec = (uint64_t)(rec0[(size_t)cand[k] + 1] - rec0[(size_t)cand[k]]); c.text_off = (uint64_t)c_off[2 * k]; memcpy(w, &c, sizeof(c)); w += sizeof(c); }
				if (bh.text_bytes) memcpy(w, ctext, (size_t)bh.text_bytes);
				w += bh.text_bytes;
				{	/* the record blob, copied by a few threads */
					const size_t nb = (size_t)bh.bam_bytes; const int T = (int)std::max<size_t>(1, std::min<size_t>(8, nb >> 24));
					std::vector<std::thread> th;
					for (int k = 0; k < T; ++k) th.emplace_back([=]() { const size_t a = nb * (size_t)k / (size_t)T, e = nb * (size_t)(k + 1) / (size_t)T; memcpy(w + a, bam + a, e - a); });
					for (std::thread &x : th) x.join();
				}
				ssg_free(bam); ssg_free(ctext);
			}
			busy += wall() - t0;
			const ssg_pestat_t *pp = ssg_pe_pes(B->res);
			if (se) fprintf(stderr, "[bwa] processed %d single-end reads on %s device %d\n", n, ssg_backend(), B->dev);
			else fprintf(stderr, "[bwa] processed %d reads in %d upstream batch(es) on %s device %d; FR insert (first batch): failed=%d low=%d high=%d avg=%.2f std=%.2f\n",
			        n, B->n_batches, ssg_backend(), B->dev, pp[1].failed, pp[1].low, pp[1].high, pp[1].avg, pp[1].std);
			ssg_pe_result_free(B->res);
			emit(B->seqno, t);
		}
		std::lock_guard<std::mutex> l(ord.mu); tm_fmt += busy;
	};
	{	std::vector<std::thread> t_fmt;
		for (int k = 1; k < n_fmt; ++k) t_fmt.emplace_back(fmt_loop);
		fmt_loop();
		for (std::thread &x : t_fmt) x.join();
		for (auto &kv : ord.pend) {   /* only after a failure: batches behind a lost one */
			text_t &t = kv.second;
			if (t.seg) { t.seg->reset(); unlink(t.seg_path->c_str()); delete t.seg; delete t.seg_path; } else ssg_free(t.p);
		}
	}
	if (fused && !fail) { text_t t; t.len = 0; t.frame = FU_END; t.seg = 0; t.seg_path = 0; t.p = (char*)malloc(1); to_write.push(t); }
	to_write.close(); t_write.join();
	t_asm.join(); for (std::thread &x : t_gpu) x.join();
	if (t_warm.joinable()) t_warm.join();
	ssg_stamp("bwa", "output_closed");
	fprintf(stderr, "[bwa] wall: index load %.2f s, reads -> %s %.2f s\n", t_loaded - t_start, fused ? "BAM records (fused)" : "SAM", wall() - t_loaded);
	{ double g = 0; for (double x : tm_gpu) g += x; fprintf(stderr, "[bwa] stage busy time: assemble %.2f s, device call %.2f s, format %.2f s\n", tm_asm, g, tm_fmt); }
	if (n_dev > 1) for (int g = 0; g < n_dev; ++g) fprintf(stderr, "[bwa] device %d: %ld calls, %.2f s busy\n", g, calls[(size_t)g], tm_gpu[(size_t)g]);
	if (getenv("SSG_BWA_PROF")) {   /* per-kernel device time of the whole run (HIP events; the profiling was switched on before the first call) */
		const char *nm[256]; double ms[256]; long cnt[256];
		const int n = std::min(ssg_prof_get(256, nm, ms, cnt), 256);
		std::vector<int> o(n); for (int i = 0; i < n; ++i) o[i] = i;
		std::sort(o.begin(), o.end(), [&](int a, int b) { return ms[a] > ms[b]; });
		for (int i = 0; i < n && i < 24; ++i) fprintf(stderr, "[bwa] kernel %-28s %9.1f ms in %ld launches\n", nm[o[i]], ms[o[i]], cnt[o[i]]);
	}
	for (fq_feed_t *f : { feed1.get(), feed2.get(), fb1.get(), fb2.get() }) if (f) { std::shared_ptr<fq_block_t> drop; while (f->ch.pop(drop)) {} f->th.join(); }   /* let the readers finish after an error */
	fb1.reset(); fb2.reset(); rawf.reset();   /* (the fall-back parsers read from the device-text path's decoders) */
	gzclose(fp1); if (fp2) gzclose(fp2);
	for (int g = 0; g < n_dev; ++g) { (void)ssg_set_device(g); ssg_index_destroy(idxs[(size_t)g]); }
	if (fail) rk_mark_failed("bwa");
	ssg_stamp("bwa", "end");
	return fail ? 1 : 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: bwa <index|mem> ...  (libssgpu %s, %s)\n", ssg_version(), ssg_backend()); return 1; }
	if (!strcmp(argv[1], "index")) return main_index(argc - 1, argv + 1);
	if (!strcmp(argv[1], "mem")) return ssg_fast_exit(main_mem(argc - 1, argv + 1));
	fprintf(stderr, "[bwa] unknown command %s\n", argv[1]);
	return 1;
}
