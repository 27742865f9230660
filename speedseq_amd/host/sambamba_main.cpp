/*
 * sambamba_main.cpp -- the `sambamba` executable speedseq.config names (reference bin/speedseq.config:9), with the
 * sub-commands the reference's scripts issue:
 *   view -S -f bam [-l N] /dev/stdin          SAM text -> BAM on stdout               bin/speedseq:426,433,440,447
 *   view -H <in.bam>                          header text                             bin/speedseq:682,1032,1179
 *   sort -t N -m XG --tmpdir=DIR -o out <in>  coordinate sort                         bin/speedseq:427,431,434,1950
 *   index <in.bam>                            <in.bam>.bai                            bin/speedseq:486-494
 *   merge -t N out.bam in1.bam in2.bam ...    merge of coordinate-sorted files        bin/speedseq:2002-2004
 * Formats and orders follow the in-tree samtools / htslib 1.3.1 (bamio.h cites the lines).  Text parsing, BGZF (de)compression
 * and the gather of sorted records run on host threads; the coordinate sort itself -- a stable sort of the 64-bit keys
 * tid<<32 | (pos+1)<<1 | reverse -- runs on the MI355X through libssgpu (ssg_sort_u64_perm), chunk by chunk when the input
 * exceeds the -m budget (chunks are spilled to --tmpdir and merged; ties keep input order, as samtools' merge of sorted blocks).
 */
#include <queue>
#include <memory>
#include <fcntl.h>
#include <sys/stat.h>
#include "bamio.h"
#include "fastq.h"   /* chan_t */
#include "fused.h"
#include "xchg.h"
#include "ranks.h"
#include <atomic>
#include <mutex>
#include <condition_variable>
#include "../../include/ssgpu.h"

#include <time.h>
static double wall() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static bool dbg() { static int d = -1; if (d < 0) d = (getenv("SSG_DEBUG") || getenv("SSG_SORT_LOG")) ? 1 : 0; return d != 0; }
static int hw_threads() { unsigned n = std::thread::hardware_concurrency(); return n ? (int)std::min(n, 32u) : 4; }
static void die(const std::string &m) { fprintf(stderr, "[sambamba] %s\n", m.c_str()); rk_mark_failed("sambamba"); exit(1); }   /* rank mode: the other ranks must not wait for this one */
static int open_in(const char *p) { if (!strcmp(p, "/dev/stdin") || !strcmp(p, "-")) return 0; int fd = open(p, O_RDONLY); if (fd < 0) die(std::string("cannot open ") + p); return fd; }

/* ---------------- view ---------------- */
static int cmd_view(int argc, char **argv)
{
	int level = -1, threads = hw_threads(); bool hdr_only = false, sam_in = false; const char *in = 0, *fmt = "sam";
	for (int i = 0; i < argc; ++i) {
		const char *a = argv[i];
		if (!strcmp(a, "-S")) sam_in = true;
		else if (!strcmp(a, "-H")) hdr_only = true;
		else if (!strcmp(a, "-h")) ;
		else if (!strcmp(a, "-f") && i + 1 < argc) fmt = argv[++i];
		else if (!strcmp(a, "-l") && i + 1 < argc) level = atoi(argv[++i]);
		else if (!strcmp(a, "-t") && i + 1 < argc) threads = atoi(argv[++i]);
		else if (a[0] == '-' && a[1] && strcmp(a, "-")) die(std::string("view: unsupported option ") + a);
		else in = a;
	}
	if (!in) die("usage: sambamba view [-S] [-f bam] [-l N] [-H] <in>");
	const int fd = open_in(in);
	if (hdr_only) {
		bgzf_in_t bi(fd, 1); bam_hdr_t h;
		if (!hdr_read(bi, h)) die("view -H: not a BAM file");
		io_write_all(1, h.text.data(), h.text.size());
		return 0;
	}
	if (!sam_in || strcmp(fmt, "bam")) die("view: only `-S -f bam` (SAM text to BAM) and `-H` are supported");
	/* fused stream (fused.h): the records already are BAM records; they pass through to `sambamba sort` untouched */
	char first[8]; size_t n_first = 0;
	while (n_first < 8) { ssize_t r = read(fd, first + n_first, 8 - n_first); if (r < 0) { if (errno == EINTR) continue; die("view: read error"); } if (r == 0) break; n_first += (size_t)r; }
	if (n_first == 8 && !memcmp(first, FU_MAGIC, 8)) {
		io_write_all(1, first, 8);
		std::vector<char> b((size_t)4 << 20);
		for (;;) { ssize_t r = read(fd, b.data(), b.size()); if (r < 0) { if (errno == EINTR) continue; die("view: read error"); } if (r == 0) break; io_write_all(1, b.data(), (size_t)r); }
		return 0;
	}
	bgzf_out_t out(1, level, threads);
	bam_hdr_t h; bool hdr_done = false;
	/* a reader thread cuts stdin into pieces of whole lines (the pipe from samblaster is read while the previous piece is encoded) */
	struct piece_t { std::vector<char> buf; size_t end; };
	chan_t<std::unique_ptr<piece_t> > ch(2);
	std::thread reader([&]() {
		std::vector<char> carry(first, first + n_first); bool eof = false;
		while (!eof) {
			std::unique_ptr<piece_t> P(new piece_t());
			P->buf.resize((size_t)48 << 20);
			size_t have = carry.size();
			if (have > P->buf.size() / 2) P->buf.resize(have * 2 + ((size_t)1 << 20));
			memcpy(P->buf.data(), carry.data(), have); carry.clear();
			while (!eof && have < P->buf.size()) {
				ssize_t r = read(fd, P->buf.data() + have, P->buf.size() - have);
				if (r < 0) { if (errno == EINTR) continue; die("view: read error"); }
				if (r == 0) { eof = true; break; }
				have += (size_t)r;
			}
			if (eof && have && P->buf[have - 1] != '\n') { if (have == P->buf.size()) P->buf.resize(have + 1); P->buf[have++] = '\n'; }
			size_t end = have;
			while (end > 0 && P->buf[end - 1] != '\n') --end;       /* complete lines only; the rest starts the next piece */
			carry.assign(P->buf.data() + end, P->buf.data() + have);
			P->end = end;
			if (end) ch.push(std::move(P));
		}
		ch.close();
	});
	std::unique_ptr<piece_t> P;
	while (ch.pop(P)) {
		const std::vector<char> &buf = P->buf; const size_t end = P->end;
		size_t p = 0;
		while (!hdr_done && p < end) {
			if (buf[p] != '@') { hdr_done = true; hdr_from_text(h); hdr_write(out, h); break; }
			const char *nl = (const char*)memchr(buf.data() + p, '\n', end - p);
			h.text.append(buf.data() + p, (size_t)(nl - (buf.data() + p)) + 1);
			p = (size_t)(nl - buf.data()) + 1;
		}
		if (hdr_done && p < end) {
			/* threads take byte ranges cut at line ends and encode the lines inside */
			const size_t T = (size_t)std::max(1, threads);
			std::vector<size_t> cutp(T + 1, end); cutp[0] = p;
			for (size_t t = 1; t < T; ++t) {
				size_t q = p + (end - p) * t / T;
				if (q < cutp[t - 1]) q = cutp[t - 1];
				const char *nl = q < end ? (const char*)memchr(buf.data() + q, '\n', end - q) : 0;
				cutp[t] = nl ? (size_t)(nl - buf.data()) + 1 : end;
			}
			std::vector<std::vector<uint8_t> > enc(T); std::vector<std::string> errs(T);
			parallel_for((int)T, T, [&](size_t a, size_t b, int) {
				for (size_t t = a; t < b; ++t) {
					enc[t].reserve((cutp[t + 1] - cutp[t]) * 9 / 10 + 1024);
					for (size_t q = cutp[t]; q < cutp[t + 1] && errs[t].empty(); ) {
						const char *s = buf.data() + q, *nl = (const char*)memchr(s, '\n', cutp[t + 1] - q), *e = nl;
						q = (size_t)(nl - buf.data()) + 1;
						while (e > s && e[-1] == '\r') --e;
						if (e == s) continue;
						std::string er;
						if (sam_line_to_bam(s, e, h, enc[t], er)) errs[t] = er + ": " + std::string(s, std::min<size_t>(80, (size_t)(e - s)));
					}
				}
			});
			for (auto &er : errs) if (!er.empty()) die("view: malformed SAM line (" + er + ")");
			for (auto &v : enc) out.put(v.data(), v.size());        /* records may straddle BGZF blocks (bgzf_write), as in any BAM stream */
		}
	}
	reader.join();
	if (!hdr_done) { hdr_from_text(h); hdr_write(out, h); }
	out.finish();
	return 0;
}

/* ---------------- sort ---------------- */
static void change_so(std::string &text, const char *so)
{	/* samtools bam_sort.c change_SO: @HD ... SO:<so> (added when there is no @HD line) */
	if (text.size() > 3 && !text.compare(0, 3, "@HD")) {
		size_t nl = text.find('\n'); if (nl == std::string::npos) return;
		size_t q = text.find("\tSO:");
		if (q != std::string::npos && q < nl) {
			size_t e = q + 4; while (e < text.size() && text[e] != '\n' && text[e] != '\t') ++e;
			if (text.compare(q + 4, e - q - 4, so) == 0) return;
			text.replace(q, e - q, std::string("\tSO:") + so);
		} else text.insert(nl, std::string("\tSO:") + so);
	} else text = std::string("@HD\tVN:1.3\tSO:") + so + "\n" + text;
}

/* records of the input, held in the chunks they arrived in (a fused frame, or ~64 MB assembled from the BGZF stream): nothing is
 * copied or re-allocated while the input streams in; a record is (chunk << 40 | offset of its block_size word) */
struct rec_store_t {
	std::vector<fu_buf_t> chunk; std::vector<size_t> chunk_len;        /* heap memory, or the mapped segment a fused frame arrived in (fused.h) */
	std::vector<uint64_t> loc, key, ord; uint64_t bytes;      /* ord (rank mode only): input ordinal of a record over all ranks, (batch << 28 | index in the batch) */
	rec_store_t() : bytes(0) {}
	const uint8_t *rec(size_t i) const { return chunk[(size_t)(loc[i] >> 40)].p + (loc[i] & (((uint64_t)1 << 40) - 1)); }
	void clear() { chunk.clear(); chunk_len.clear(); loc.clear(); key.clear(); ord.clear(); bytes = 0; }
	/* index the whole records of chunk c[0..len) */
	bool add_chunk(fu_buf_t c, size_t len, uint64_t ord_base = ~(uint64_t)0)
	{
		const uint64_t id = chunk.size(); const uint8_t *p = c.p; size_t o = 0; uint64_t k = 0;
		while (o + 4 <= len) { uint32_t bs; memcpy(&bs, p + o, 4); if (o + 4 + (size_t)bs > len || bs < 32) return false; loc.push_back(id << 40 | (uint64_t)o); key.push_back(bam_sort_key(p + o + 4)); if (ord_base != ~(uint64_t)0) ord.push_back(ord_base + k++); o += 4 + (size_t)bs; }
		if (o != len) return false;
		chunk.push_back(std::move(c)); chunk_len.push_back(len); bytes += len;
		return true;
	}
};

/* the whole records of a received block with the ordinals that came with them (rank mode's exchange) */
static bool add_chunk_ords(rec_store_t &S, fu_buf_t c, size_t len, const uint64_t *ords, size_t n)
{
	const size_t n0 = S.key.size();
	if (!S.add_chunk(std::move(c), len)) return false;
	if (S.key.size() - n0 != n) return false;
	S.ord.resize(n0);
	S.ord.insert(S.ord.end(), ords, ords + n);
	return true;
}

static void gpu_perm(const rec_store_t &S, std::vector<uint32_t> &perm)
{
	perm.resize(S.key.size());
	if (S.key.empty()) return;
	if (ssg_sort_u64_perm(S.key.data(), (int64_t)S.key.size(), perm.data())) die(std::string("sort: ") + ssg_last_error());
}

/* the sorted records as a BGZF stream.  Blocks are cut exactly as bgzf_out_t::record cuts them (a record does not straddle blocks
 * unless it is larger than one); the gather of a block's records and its deflate run on the thread pool, groups of blocks are
 * written in order by the calling thread while later groups are still being compressed. */
/* <bam>.bai.ssg: "size mtime_s mtime_ns crc32(bai)" of a BAM and the index `sambamba sort` wrote with it */
static bool bai_note_make(const std::string &bam, std::string &note)
{
	struct stat sb; if (stat(bam.c_str(), &sb) != 0) return false;
	FILE *f = fopen((bam + ".bai").c_str(), "rb"); if (!f) return false;
	uLong crc = crc32(0L, Z_NULL, 0); uint8_t buf[65536]; size_t k;
	while ((k = fread(buf, 1, sizeof(buf), f)) > 0) crc = crc32(crc, buf, (uInt)k);
	fclose(f);
	char t[128]; snprintf(t, sizeof(t), "%lld %lld %ld %lu\n", (long long)sb.st_size, (long long)sb.st_mtim.tv_sec, (long)sb.st_mtim.tv_nsec, (unsigned long)crc);
	note = t; return true;
}
static void bai_note_write(const std::string &bam)
{
	std::string note; if (!bai_note_make(bam, note)) return;
	FILE *f = fopen((bam + ".bai.ssg").c_str(), "w"); if (!f) return;
	fputs(note.c_str(), f); fclose(f);
}

/* bai_path != NULL (the final file, a regular file): the .bai is written as well -- every field `sambamba index` would read back from the
 * file is in memory here -- with a note next to it (<bai>.ssg: size and mtime of the BAM, CRC of the index) that lets `sambamba index`,
 * which the reference runs right after the sort (bin/speedseq:491-495), recognise the pair as current instead of inflating the BAM again */
/* force_at != NULL (a sorted run on its way to the temporary directory, cmd_sort): no header; a block starts at each of these record
 * indices (ascending, <= n) and force_off receives the file offset of that block -- the run's segments, one per range of the genome */
/* seg != NULL (the merge of sorted runs, merge_runs): the file is written by several calls, one per stretch of the genome; the first writes the
 * header, the last the end-of-file block and the index, and the state that crosses calls lives in *seg */
struct seg_out_t {
	bool first, last;                                  /* set by the caller for each call */
	uint64_t coff;                                     /* file offset of the next block */
	std::unique_ptr<bai_t> idx; bool idx_ok;
	bool no_index;                                     /* the output cannot be sought: no index */
	uint64_t hdr_end;                                  /* file offset of the first block after the header */
	int ent_fd;                                        /* >= 0 (rank mode): every record's index entry + virtual offset in this file goes here, 24 bytes each (cmd_index_parts) */
	bool pending; int32_t p_tid, p_pos, p_end; bool p_mapped;   /* the last record of the previous stretch: its entry closes at the first record of the next */
	seg_out_t() : first(true), last(false), coff(0), idx_ok(true), no_index(false), hdr_end(0), ent_fd(-1), pending(false), p_tid(0), p_pos(0), p_end(0), p_mapped(false) {}
};
static void write_sorted(const rec_store_t &S, const std::vector<uint32_t> &perm, const bam_hdr_t &h, int fd, int level, int threads, const char *bai_path = 0,
                         const std::vector<size_t> *force_at = 0, std::vector<uint64_t> *force_off = 0, seg_out_t *seg = 0, std::vector<uint64_t> *force_uoff = 0)
{
	const bool opens = !seg || seg->first, closes = !seg || seg->last;
	if (!force_at && opens) { bgzf_out_t out(fd, level, threads); hdr_write(out, h); out.drain(true); }
	off_t hdr_end = force_at ? (off_t)0 : !opens ? (off_t)seg->coff : (bai_path || seg) ? lseek(fd, 0, SEEK_CUR) : (off_t)-1;
	if (hdr_end < 0) { bai_path = 0; if (seg) { seg->no_index = true; hdr_end = 0; } }   /* a pipe: no offsets, no index (the file offsets kept in *seg then only count from the first record) */
	if (seg && seg->no_index) bai_path = 0;
	if (seg && opens) seg->hdr_end = (uint64_t)hdr_end;
	const bool want_off = bai_path || force_at || seg;
	const size_t n = perm.size();
	const int lvl = level < 0 ? 6 : level;
	const double tw0 = wall();
	/* virtual byte offsets of the sorted stream and the block cuts */
	std::vector<uint64_t> cum(n + 1); cum[0] = 0;
	parallel_for((int)std::min<size_t>((size_t)std::max(1, threads), n / 65536 + 1), n, [&](size_t a, size_t b, int) { for (size_t i = a; i < b; ++i) { uint32_t bs; memcpy(&bs, S.rec(perm[i]), 4); cum[i + 1] = 4 + (uint64_t)bs; } });
	for (size_t i = 0; i < n; ++i) cum[i + 1] += cum[i];
	std::vector<uint64_t> cut; cut.push_back(0);
	std::vector<size_t> force_blk(force_at ? force_at->size() : 0, 0);   /* block that starts at each forced index */
	{	uint64_t open = 0;   /* start of the open block */
		size_t fk = 0;
		for (size_t i = 0; i < n; ++i) {
			const uint64_t sz = cum[i + 1] - cum[i];
			if (force_at && fk < force_at->size() && (*force_at)[fk] == i) {
				if (cum[i] > open) { cut.push_back(cum[i]); open = cum[i]; }
				while (fk < force_at->size() && (*force_at)[fk] == i) force_blk[fk++] = cut.size() - 1;
			}
			if (cum[i] > open && cum[i] - open + sz > BGZF_MAX_PAYLOAD) { cut.push_back(cum[i]); open = cum[i]; }
			while (cum[i + 1] - open >= BGZF_MAX_PAYLOAD) { open += BGZF_MAX_PAYLOAD; cut.push_back(open); }   /* a record larger than a block fills whole blocks */
		}
		if (cum[n] > cut.back()) cut.push_back(cum[n]);
	}
	const double tw1 = wall();
	const size_t nb = cut.size() - 1, GRP = 16, ng = (nb + GRP - 1) / GRP;   /* 1 MB of records per work item: a sort of a few hundred blocks still spreads over the pool, and the last items of a large one end together */
	/* Hand-over between the pool, the writer and the index thread is by atomics and short sleeps, not by one mutex + condition variable: with
	 * 256 workers waiting on one condition every notify_all of the writer woke them all to fight for the mutex (round 4's first timers:
	 * 1.98 s of deflate per worker inside 14.5 s of wall, 11.5 s of it `held back'); nobody here waits for long, so polling costs nothing. */
	struct grp_t { std::vector<uint8_t> bytes; std::vector<uint32_t> bsz; };
	std::vector<uint64_t> blk_coff(want_off ? nb + 1 : 0);      /* file offset of every block */
	std::vector<grp_t> grp(ng);
	std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[ng ? ng : 1]);
	for (size_t g = 0; g < ng; ++g) done[g].store(0, std::memory_order_relaxed);
	std::atomic<size_t> next_write(0), next_grp(0);
	/* blocks deflated on the device: the final file of a sort at a compressing level, when there is a device (SSG_BGZF_DEVICE=0 keeps zlib on the
	 * host's threads; the host emulation of the kernels only does it on request, it is slow) */
	const char *const bd = getenv("SSG_BGZF_DEVICE");
	/* a run's blocks by the host's zlib pool (level 1), the device left to `bwa mem', where the host has the cores for it (32 usable ones): at 200 M pairs behind a 16-core
	 * quota both ways end at 0.95-0.97 M pairs/s -- the device way slows bwa's kernels, the host way starves its threads (profiles/r06_soak_200M*.json).  SSG_SORT_RUN_DEVICE=0 / 1 decides. */
	const char *const rde = getenv("SSG_SORT_RUN_DEVICE");
	const bool run_on_host = force_at && (rde && *rde ? atoi(rde) == 0 : ssg_usable_cores() >= 32);
	const bool use_dev = lvl != 0 && nb > 0 && !run_on_host && !(bd && !strcmp(bd, "0")) && (bd || strcmp(ssg_backend(), "emu") != 0) && ssg_device_count() > 0 && GRP * 2 <= 2048 && 2048 % GRP == 0;
	size_t DEV_BATCH = 2048; { const char *e = getenv("SSG_SORT_DEV_BATCH"); if (e && atol(e) >= (long)GRP && atol(e) <= 2048 && atol(e) % (long)GRP == 0) DEV_BATCH = (size_t)atol(e); }   /* tests: several producers on several devices for a small file */
	/* every visible device deflates (SSG_SORT_DEVICES caps them): producer t works on device t mod n_dev, lane 1 + t / n_dev -- with one device three
	 * producers on three lanes as before, with N devices at least two per device; the writer and the index thread do not care who made a block */
	int n_devs = use_dev ? std::max(1, ssg_device_count()) : 1; { const char *e = getenv("SSG_SORT_DEVICES"); if (e && atoi(e) > 0) n_devs = std::min(n_devs, atoi(e)); }
	/* (SSG_SORT_PRODUCERS overrides; six producers on the one device -- a producer gathers and checksums on the host OR deflates on the device, never both at once --
	 * were slower than three on the 16-CPU host next to the MI355X: 1.33 vs 1.15 s for 5.1 GB, profiles/r06f_literal_sort_producers.json: the gather threads are the limit) */
	int want_prod = std::min(3 * n_devs, std::max(3, 2 * n_devs)); { const char *e = getenv("SSG_SORT_PRODUCERS"); if (e && atoi(e) > 0) want_prod = std::min(7 * n_devs, atoi(e)); }
	/* a run (force_at: written while the input still arrives and `bwa mem' works on the same device) takes little of it: one producer, batches of 512 blocks -- a
	 * wave of the deflate kernel holds 22 KB of LDS, seven of them fill a CU, and three producers' batches held the extension kernels of `bwa mem' to a third of
	 * their rate for as long as a run was written (profiles/r06_soak_200M.json) */
	if (force_at && use_dev && !getenv("SSG_SORT_DEV_BATCH")) { DEV_BATCH = 512; want_prod = 1; }
	const int n_prod = use_dev ? (int)std::max<size_t>(1, std::min<size_t>((size_t)want_prod, (nb + DEV_BATCH - 1) / DEV_BATCH)) : 0;
	/* ... and on a host with many cores next to the device, some batches stay with the host's pool (zlib): batch k of the file belongs to slot k mod (producers + host
	 * slots), a fixed rule -- the file's bytes do not depend on who was faster.  SSG_SORT_HOST_BATCHES: host slots per round (2 from 96 usable cores, 1 from 48, else 0: zlib at level 6 makes 35 MB/s a core). */
	int host_slots = 0;
	if (use_dev) { const char *e = getenv("SSG_SORT_HOST_BATCHES"); const double cores = std::min((double)threads, ssg_usable_cores()); host_slots = e && *e ? std::max(0, std::min(16, atoi(e))) : cores >= 96 ? 2 : cores >= 48 ? 1 : 0; if ((nb + DEV_BATCH - 1) / DEV_BATCH <= (size_t)n_prod) host_slots = 0; }
	const int slot_round = n_prod + host_slots;
	std::atomic<bool> host_takes_all(!use_dev);
	std::atomic<int> dev_failed(0);
	auto host_group = [&](size_t g) -> bool { return host_takes_all.load(std::memory_order_relaxed) || (int)((g * GRP / DEV_BATCH) % (size_t)slot_round) >= n_prod; };
	const int n_workers = use_dev ? (host_slots ? (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), ng)) : 0)
	                              : (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), ng));   /* no more threads than work items */
	const size_t window = use_dev ? (size_t)std::max(8, 2 * n_prod) * DEV_BATCH / GRP : std::max<size_t>(64, (size_t)n_workers * 8);   /* work items compressed ahead of the writer (memory bound: ~0.3 MB each) */
	auto nap = [](int us) { std::this_thread::sleep_for(std::chrono::microseconds(us)); };
	std::atomic<long> us_gather(0), us_deflate(0), us_window(0);   /* summed over the workers (SSG_DEBUG) */
	/* payload of block bk (records cut[bk] .. cut[bk+1] of the sorted stream, a record larger than a block split over several) into dst */
	auto gather_block = [&](size_t bk, uint8_t *dst) -> size_t {
		const uint64_t v0 = cut[bk], v1 = cut[bk + 1];
		size_t i = (size_t)(std::upper_bound(cum.begin(), cum.end(), v0) - cum.begin()) - 1; size_t w = 0;
		for (uint64_t v = v0; v < v1; ++i) { const uint8_t *r = S.rec(perm[i]); const uint64_t a = v - cum[i], e = std::min(cum[i + 1], v1) - cum[i]; memcpy(dst + w, r + a, (size_t)(e - a)); w += (size_t)(e - a); v = cum[i] + e; }
		return w;
	};
	auto worker = [&]() {
		std::vector<uint8_t> payload(BGZF_MAX_PAYLOAD), blk(65536);
		long my_g = 0, my_d = 0, my_w = 0;
		for (;;) {
			const size_t g = next_grp.fetch_add(1);
			if (g >= ng) break;
			if (done[g].load(std::memory_order_acquire)) continue;   /* (made on the device before it failed: see the writer's fall-back) */
			if (!host_group(g)) continue;                            /* (a batch of the device's producers) */
			bool give_up = false;   /* (the device failed while this thread waited: the writer wants every thread joined before it starts the host-only pool, which makes this group then) */
			if (g >= next_write.load(std::memory_order_acquire) + window) {
				const double t0 = wall();
				while (g >= next_write.load(std::memory_order_acquire) + window && !(give_up = dev_failed.load() && !host_takes_all.load())) nap(200);
				my_w += (long)((wall() - t0) * 1e6);
			}
			if (give_up) break;
			std::vector<uint8_t> ob; ob.reserve(GRP * (lvl ? 24576 : 65536));
			std::vector<uint32_t> bsz;
			for (size_t bk = g * GRP; bk < std::min(nb, (g + 1) * GRP); ++bk) {
				const double t0 = wall();
				const size_t w = gather_block(bk, payload.data());
				const double t1 = wall();
				const size_t k = bgzf_make_block(payload.data(), w, lvl, blk.data());
				ob.insert(ob.end(), blk.data(), blk.data() + k); bsz.push_back((uint32_t)k);
				my_g += (long)((t1 - t0) * 1e6); my_d += (long)((wall() - t1) * 1e6);
			}
			grp[g].bytes.swap(ob); grp[g].bsz.swap(bsz);
			done[g].store(1, std::memory_order_release);
		}
		us_gather += my_g; us_deflate += my_d; us_window += my_w;
	};
	/* ... or the blocks are deflated on the device (ssg_bgzf_deflate, k_bgzf.h): a few producer threads, each with a stream of its own, take
	 * batches of 2048 blocks in turn -- gather into page-locked memory + CRC-32 by host threads, deflate on the GPU, framing (BGZF header,
	 * CRC, ISIZE) -- and hand the groups to the same in-order writer.  The host's cores, which the deflate of a whole genome's records kept busy
	 * for longer than the alignment took, only copy and checksum. */
	std::atomic<long> dev_batches(0); const long fail_after = getenv("SSG_BGZF_FAIL_AFTER") ? atol(getenv("SSG_BGZF_FAIL_AFTER")) : -1;   /* tests: the device "fails" from its n-th batch on */
	auto producer = [&](int t) {
		if (ssg_set_device(t % n_devs) || ssg_set_lane(1 + (t / n_devs) % 7)) { dev_failed = 1; return; }
		const int gth = std::max(1, threads / std::max(1, n_prod));
		/* two slots: a helper thread gathers and checksums batch k + 1 on the host while this one has batch k deflated on the device (until round 6 a producer did one
		 * after the other, and the device waited whenever all producers were gathering: 0.74 s of 1.15 s for 5.1 GB, profiles/r06f_literal_sort_producers.json) */
		struct slot_t { uint8_t *P; std::vector<uint64_t> rel; std::vector<uint32_t> crc; size_t b0, n_b; std::atomic<int> state; slot_t() : P(0), b0(0), n_b(0), state(0) {} };   /* state: 0 free, 1 gathered, 2 no more batches */
		slot_t slot[2];
		for (slot_t &sl : slot) { sl.P = (uint8_t*)ssg_host_alloc(DEV_BATCH * BGZF_MAX_PAYLOAD + 64); sl.rel.resize(DEV_BATCH + 1); sl.crc.resize(DEV_BATCH); }
		uint8_t *O = (uint8_t*)ssg_host_alloc(DEV_BATCH * (BGZF_MAX_PAYLOAD + 5) + 64);
		std::vector<uint64_t> off(DEV_BATCH + 1);
		std::atomic<long> my_g(0); long my_d = 0; std::atomic<long> my_w(0);
		const bool mem_ok = slot[0].P && slot[1].P && O;
		std::thread gatherer([&]() {
			int k = 0;
			for (size_t b0 = (size_t)t * DEV_BATCH; b0 < nb && mem_ok && !dev_failed.load(); b0 += (size_t)slot_round * DEV_BATCH, k ^= 1) {
				slot_t &sl = slot[k];
				while (sl.state.load(std::memory_order_acquire) != 0 && !dev_failed.load()) nap(100);
				const size_t b1 = std::min(nb, b0 + DEV_BATCH), n_b = b1 - b0, g0 = b0 / GRP;
				if (g0 >= next_write.load(std::memory_order_acquire) + window) { const double t0 = wall(); while (g0 >= next_write.load(std::memory_order_acquire) + window && !dev_failed.load()) nap(200); my_w += (long)((wall() - t0) * 1e6); }
				if (dev_failed.load()) break;   /* (the writer is waiting for the producers to stop before it hands the rest to the host's pool) */
				const double t0 = wall();
				for (size_t i = 0; i <= n_b; ++i) sl.rel[i] = cut[b0 + i] - cut[b0];
				parallel_for((int)std::min<size_t>((size_t)gth, n_b / 8 + 1), n_b, [&](size_t a, size_t e, int) {
					for (size_t i = a; i < e; ++i) { const size_t w = gather_block(b0 + i, sl.P + sl.rel[i]); sl.crc[i] = (uint32_t)crc32(crc32(0L, Z_NULL, 0), sl.P + sl.rel[i], (uInt)w); }
				});
				my_g += (long)((wall() - t0) * 1e6);
				sl.b0 = b0; sl.n_b = n_b;
				sl.state.store(1, std::memory_order_release);
			}
			for (int j = 0; j < 2; ++j, k ^= 1) {   /* the end: in the slot order the consumer follows */
				slot_t &sl = slot[k];
				while (sl.state.load(std::memory_order_acquire) != 0 && !dev_failed.load()) nap(100);
				if (dev_failed.load()) break;
				sl.state.store(2, std::memory_order_release);
			}
		});
		for (int k = 0; mem_ok; k ^= 1) {
			slot_t &sl = slot[k];
			int st;
			while ((st = sl.state.load(std::memory_order_acquire)) == 0 && !dev_failed.load()) nap(50);
			if (st != 1) break;
			const size_t b0 = sl.b0, n_b = sl.n_b, b1 = b0 + n_b, g0 = b0 / GRP, g1 = (b1 + GRP - 1) / GRP;
			const double t1 = wall();
			if ((fail_after >= 0 && dev_batches.fetch_add(1) >= fail_after) || ssg_bgzf_deflate(sl.P, sl.rel.data(), (long)n_b, O, (uint64_t)DEV_BATCH * (BGZF_MAX_PAYLOAD + 5) + 64, off.data())) {
				fprintf(stderr, "[sambamba] sort: BGZF deflate on the device failed: %s\n", fail_after >= 0 ? "(SSG_BGZF_FAIL_AFTER: test)" : ssg_last_error()); dev_failed = 1; break; }
			const double t2 = wall();
			parallel_for((int)std::min<size_t>((size_t)std::min(gth, 8), g1 - g0), g1 - g0, [&](size_t a, size_t e, int) {
				static const uint8_t hdr[16] = { 0x1f,0x8b,0x08,0x04,0,0,0,0,0,0xff,0x06,0,0x42,0x43,0x02,0 };
				for (size_t g = g0 + a; g < g0 + e; ++g) {
					std::vector<uint8_t> ob; std::vector<uint32_t> bsz;
					for (size_t bk = g * GRP; bk < std::min(nb, (g + 1) * GRP); ++bk) {
						const size_t i = bk - b0, clen = (size_t)(off[i + 1] - off[i]), bsize = 18 + clen + 8, at = ob.size();
						ob.resize(at + bsize);
						uint8_t *d = ob.data() + at;
						memcpy(d, hdr, 16); d[16] = (uint8_t)((bsize - 1) & 0xff); d[17] = (uint8_t)((bsize - 1) >> 8);
						memcpy(d + 18, O + off[i], clen);
						const uint32_t isz = (uint32_t)(sl.rel[i + 1] - sl.rel[i]);
						memcpy(d + 18 + clen, &sl.crc[i], 4); memcpy(d + 18 + clen + 4, &isz, 4);
						bsz.push_back((uint32_t)bsize);
					}
					grp[g].bytes.swap(ob); grp[g].bsz.swap(bsz);
					done[g].store(1, std::memory_order_release);
				}
			});
			my_d += (long)((t2 - t1) * 1e6);
			sl.state.store(0, std::memory_order_release);
		}
		if (!mem_ok) dev_failed = 1;
		gatherer.join();
		for (slot_t &sl : slot) ssg_host_free(sl.P);
		ssg_host_free(O);
		us_gather += my_g.load(); us_deflate += my_d; us_window += my_w.load();
	};
	/* the index `sambamba index` would make of this file (cmd_index below: same bai_t calls, same virtual offsets), built by a thread of
	 * its own that follows the writer: a record's virtual offset is known as soon as the group holding its block has its file offset */
	struct ent_t { int32_t tid, pos, end; uint8_t mapped; };
	const bool want_ent = bai_path || (seg && seg->ent_fd >= 0);
	std::vector<ent_t> ent(want_ent ? n : 0);
	/* the record view the index needs (position, end, mapped): made next to the producers' first batches, not before them (0.19 s of 16 M records during which the device waited) */
	auto make_ent = [&]() { if (want_ent) parallel_for((int)std::min<size_t>((size_t)std::max(1, threads / 2), n / 65536 + 1), n, [&](size_t a, size_t b, int) {
		for (size_t i = a; i < b; ++i) { const uint8_t *r = S.rec(perm[i]) + 4; bam_core_t c; memcpy(&c, r, 32); ent[i].tid = c.tid; ent[i].pos = c.pos; ent[i].end = bam_endpos(r); ent[i].mapped = !((c.flag_nc >> 16) & 4); }
	}); };
	std::thread t_ent; std::atomic<bool> ent_ready(false);
	if (bai_path && want_ent) t_ent = std::thread([&]() { make_ent(); ent_ready.store(true, std::memory_order_release); }); else { make_ent(); ent_ready.store(true); }
	struct ent_join_t { std::thread &t; ~ent_join_t() { if (t.joinable()) t.join(); } } ent_join = { t_ent };
	std::atomic<size_t> blocks_placed(0);                       /* blk_coff[0 .. blocks_placed) are final */
	std::atomic<uint64_t> file_end_v(0);                        /* virtual offset of the end of the file, 0 until the last block is placed */
	bai_t idx_own((int)h.names.size(), 0); bool idx_ok = seg ? seg->idx_ok : true; double t_idx_wait = 0;
	std::thread t_idx;
	if (bai_path) t_idx = std::thread([&]() {
		while (!ent_ready.load(std::memory_order_acquire)) nap(100);
		size_t bk = 0;
		auto wait_blocks = [&](size_t need) { if (blocks_placed.load(std::memory_order_acquire) >= need) return; const double t0 = wall(); while (blocks_placed.load(std::memory_order_acquire) < need) nap(100); t_idx_wait += wall() - t0; };
		auto voff = [&](size_t i) -> uint64_t { while (cut[bk + 1] <= cum[i]) ++bk; wait_blocks(bk + 1); return blk_coff[bk] << 16 | (cum[i] - cut[bk]); };
		auto end_of_file = [&]() -> uint64_t { wait_blocks(nb + 1); return file_end_v.load(); };
		if (!idx_ok) return;
		if (!n && !closes) return;                                     /* a stretch without records: nothing to say yet */
		bai_t *ix = &idx_own;
		if (!seg) idx_own = bai_t((int)h.names.size(), n ? voff(0) : end_of_file());
		else { if (!seg->idx) seg->idx.reset(new bai_t((int)h.names.size(), n ? voff(0) : end_of_file())); ix = seg->idx.get(); }
		if (seg && seg->pending && n) { if (ix->push(seg->p_tid, seg->p_pos, seg->p_end, voff(0), seg->p_mapped) < 0) { idx_ok = false; return; } seg->pending = false; }
		for (size_t i = 0; i < n; ++i) {
			if (i + 1 == n && !closes) { seg->pending = true; seg->p_tid = ent[i].tid; seg->p_pos = ent[i].pos; seg->p_end = ent[i].end; seg->p_mapped = ent[i].mapped != 0; break; }
			const uint64_t after = i + 1 < n ? voff(i + 1) : end_of_file();
			if (ix->push(ent[i].tid, ent[i].pos, ent[i].end, after, ent[i].mapped != 0) < 0) { idx_ok = false; break; }   /* cannot happen on a sorted stream; `sambamba index` will say so */
		}
		if (idx_ok && closes && seg && seg->pending) { if (ix->push(seg->p_tid, seg->p_pos, seg->p_end, end_of_file(), seg->p_mapped) < 0) idx_ok = false; seg->pending = false; }
		if (idx_ok && closes) ix->finish(end_of_file());
	});
	std::vector<std::thread> th;
	const double tw_spawn0 = wall();
	for (int t = 0; t < n_workers; ++t) th.emplace_back(worker);
	for (int t = 0; t < n_prod; ++t) th.emplace_back(producer, t);
	uint64_t coff = (uint64_t)(hdr_end > 0 ? hdr_end : 0);
	bool fell_back = false;
	double t_wr_wait = 0, t_wr_io = 0; const double tw_spawn = wall();
	for (size_t g = 0; g < ng; ++g) {
		if (!done[g].load(std::memory_order_acquire)) {
			const double t0 = wall();
			while (!done[g].load(std::memory_order_acquire)) {
				if (dev_failed.load() && !fell_back) {   /* a device that cannot be set up, runs out of memory or loses a kernel does not end the sort: the producers stop, the host's pool (zlib, as SSG_BGZF_DEVICE=0) makes the groups that are not there yet */
					for (auto &x : th) x.join();
					th.clear(); fell_back = true; host_takes_all = true; next_grp = 0;
					fprintf(stderr, "[sambamba] sort: compressing the rest of the output on the host (zlib level %d)\n", lvl);
					const int nw = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), ng));
					for (int t = 0; t < nw; ++t) th.emplace_back(worker);
				}
				nap(50);
			}
			t_wr_wait += wall() - t0;
		}
		std::vector<uint8_t> ob; std::vector<uint32_t> bsz; ob.swap(grp[g].bytes); bsz.swap(grp[g].bsz);
		{ const double t0 = wall(); io_write_all(fd, ob.data(), ob.size()); t_wr_io += wall() - t0; }
		if (want_off) for (size_t k = 0; k < bsz.size(); ++k) { blk_coff[g * GRP + k] = coff; coff += bsz[k]; }
		next_write.store(g + 1, std::memory_order_release);
		if (bai_path) blocks_placed.store(std::min(nb, (g + 1) * GRP), std::memory_order_release);
	}
	for (auto &x : th) x.join();
	if (t_ent.joinable()) t_ent.join();
	if (force_at) {   /* a run: no end-of-file block; the offsets of its segments (indices at n = the end of the data) */
		blk_coff[nb] = coff;
		force_off->resize(force_at->size());
		for (size_t k = 0; k < force_at->size(); ++k) (*force_off)[k] = (*force_at)[k] >= n ? coff : blk_coff[force_blk[k]];
		if (force_uoff) { force_uoff->resize(force_at->size()); for (size_t k = 0; k < force_at->size(); ++k) (*force_uoff)[k] = cum[std::min((*force_at)[k], n)]; }
		return;
	}
	if (closes) io_write_all(fd, BGZF_EOF, 28);
	const double tw2 = wall();
	if (dbg() && !seg) fprintf(stderr, "[sambamba] sort: write: offsets and block cuts %.2f s, gather + deflate + write of %zu blocks %.2f s (record view for the index %.2f s, %d workers started in %.2f s; writer: waited %.2f s for blocks, wrote for %.2f s; "
	                   "per worker: gather %.2f s, deflate %.2f s, held back by the writer's window %.2f s)\n", tw1 - tw0, nb, tw2 - tw1, tw_spawn0 - tw1, n_workers, tw_spawn - tw_spawn0, t_wr_wait, t_wr_io,
	                   us_gather / 1e6 / std::max(1, n_workers + n_prod), us_deflate / 1e6 / std::max(1, n_workers + n_prod), us_window / 1e6 / std::max(1, n_workers + n_prod));
	if (dbg() && use_dev && !seg) fprintf(stderr, "[sambamba] sort: write: blocks deflated on %d device(s) (%d producer threads; `gather' = gather + CRC-32 on the host, `deflate' = upload + kernels + download)%s\n", n_devs, n_prod,
	                                    host_slots ? (std::string("; ") + std::to_string(host_slots) + " of every " + std::to_string(slot_round) + " batches by the host's pool (zlib)").c_str() : "");
	if (seg) seg->coff = coff;
	if (seg && seg->ent_fd >= 0 && n) {   /* rank mode: what the index of the joined file needs of this stretch */
		std::vector<uint8_t> eb(24 * n); size_t bk = 0;
		for (size_t i = 0; i < n; ++i) {
			while (cut[bk + 1] <= cum[i]) ++bk;
			const uint64_t v = blk_coff[bk] << 16 | (cum[i] - cut[bk]);
			uint8_t *d = eb.data() + 24 * i; const uint32_t m = ent[i].mapped;
			memcpy(d, &ent[i].tid, 4); memcpy(d + 4, &ent[i].pos, 4); memcpy(d + 8, &ent[i].end, 4); memcpy(d + 12, &m, 4); memcpy(d + 16, &v, 8);
		}
		io_write_all(seg->ent_fd, eb.data(), eb.size());
	}
	if (!bai_path) return;
	blk_coff[nb] = coff;
	if (closes) { file_end_v.store((coff + 28) << 16); blocks_placed.store(nb + 1, std::memory_order_release); }
	t_idx.join();
	if (seg) seg->idx_ok = idx_ok;
	if (!idx_ok || !closes) return;
	(seg ? *seg->idx : idx_own).save(bai_path);
	if (dbg() && !seg) fprintf(stderr, "[sambamba] sort: write: index finished %.2f s after the last block (its thread waited %.2f s for block offsets)\n", wall() - tw2, t_idx_wait);
}

struct merge_src_t {   /* one coordinate-sorted BAM being merged */
	int fd; std::unique_ptr<bgzf_in_t> in; bam_hdr_t h; std::vector<uint8_t> rec; uint64_t key; bool ok;
	bool next() { uint32_t bs; if (in->get(&bs, 4) != 4) { ok = false; return false; } rec.resize(4 + (size_t)bs); memcpy(rec.data(), &bs, 4); if (in->get(rec.data() + 4, bs) != bs) die("merge: truncated BAM"); key = bam_sort_key(rec.data() + 4); ok = true; return true; }
};
static void kway_merge(std::vector<merge_src_t> &src, bgzf_out_t &out)
{	/* smallest key first; equal keys in source order (samtools bam_sort.c:1347-1375 heap order) */
	typedef std::pair<uint64_t, size_t> ent_t;
	std::priority_queue<ent_t, std::vector<ent_t>, std::greater<ent_t> > pq;
	for (size_t i = 0; i < src.size(); ++i) if (src[i].next()) pq.push(ent_t(src[i].key, i));
	while (!pq.empty()) {
		const size_t i = pq.top().second; pq.pop();
		out.record(src[i].rec.data(), src[i].rec.size());
		if (src[i].next()) pq.push(ent_t(src[i].key, i));
	}
}

/* ---- sorted runs in the temporary directory (input larger than -m allows in memory) ----
 * A run is written in segments, one per range of the genome (the ranges are fixed from the header's contig lengths before the first
 * run is written), each starting at a BGZF block of its own whose file offset is kept.  The final pass then is not one k-way merge
 * over whole runs -- a single thread moving every record -- but an independent small merge per range, run by the pool, each inflating
 * only its own byte range of every run and compressing its own stretch of the output; a writer puts the stretches out in order.  Ties
 * keep input order: equal keys share a range, and within a range the earlier run wins. */
struct run_t { std::string path; int fd; std::vector<uint64_t> seg, useg, rcnt; int ord_fd; run_t() : fd(-1), ord_fd(-1) {} };   /* rcnt / ord_fd (rank mode): records before each range; the run's ordinals, 8 bytes a record */   /* seg[g] .. seg[g + 1]: the run's records of range g in the file; useg: the same in record bytes */

static void make_ranges(const bam_hdr_t &h, size_t G, std::vector<uint64_t> &lo)
{	/* lo[g] = smallest sort key of range g (equal shares of the genome's length); reads without a position sort last: the last range */
	lo.assign(1, 0);
	uint64_t L = 0; for (int32_t l : h.lens) L += (uint64_t)(l > 0 ? l : 0);
	if (G < 2 || !L) return;
	size_t tid = 0; uint64_t base = 0;
	for (size_t g = 1; g < G; ++g) {
		const uint64_t X = (uint64_t)((unsigned __int128)L * g / G);
		while (tid < h.lens.size() && base + (uint64_t)std::max(h.lens[tid], 0) <= X) { base += (uint64_t)std::max(h.lens[tid], 0); ++tid; }
		if (tid >= h.lens.size()) break;
		const uint64_t key = (uint64_t)tid << 32 | (uint64_t)(uint32_t)((X - base + 1) << 1);
		if (key > lo.back()) lo.push_back(key);
	}
}

/* The merge of the runs is the in-memory sort again, a stretch of the genome at a time: consecutive ranges whose records -- of all runs -- fit
 * a third of the budget are read back (every BGZF block of a run's byte range inflated by the pool), indexed per (run, range) piece in
 * parallel, ordered by the device's stable radix sort of the keys (the pieces of one run are in order already, earlier runs come first:
 * ties keep input order) and written through write_sorted's pipeline -- gather on the host, deflate on the device, one in-order writer, the
 * .bai following it -- while the loader thread already inflates the next stretch. */
static void load_stretch(std::vector<run_t> &runs, size_t g0, size_t g1, int threads, rec_store_t &S)
{
	S.clear();
	struct piece_t { size_t chunk, run, g; uint64_t a, b; std::vector<uint64_t> loc, key; };
	std::vector<piece_t> pieces;
	for (size_t r = 0; r < runs.size(); ++r) {
		const run_t &R = runs[r];
		const uint64_t c0 = R.seg[g0], c1 = R.seg[g1], u0 = R.useg[g0], u1 = R.useg[g1];
		if (u1 == u0) continue;
		std::vector<uint8_t> raw((size_t)(c1 - c0));
		{	/* the byte range, read by several threads (page cache or disk) */
			const size_t PIECE = (size_t)8 << 20, np = (raw.size() + PIECE - 1) / PIECE;
			parallel_for((int)std::min<size_t>((size_t)std::max(1, threads), np), np, [&](size_t a, size_t b, int) {
				for (size_t k = a; k < b; ++k) {
					const size_t lo = k * PIECE, want = std::min(PIECE, raw.size() - lo);
					for (size_t got = 0; got < want; ) { const ssize_t q = pread(R.fd, raw.data() + lo + got, want - got, (off_t)(c0 + lo + got)); if (q < 0 && errno == EINTR) continue; if (q <= 0) die("sort: cannot read a sorted run back"); got += (size_t)q; }
				}
			});
		}
		struct blk_t { size_t at, bsize, xlen; uint64_t uo; uint32_t isz; };
		std::vector<blk_t> blk; uint64_t uo = 0;
		for (size_t o = 0; o < raw.size(); ) {
			const uint8_t *b = raw.data() + o;
			if (o + 18 > raw.size() || b[0] != 0x1f || b[1] != 0x8b || !(b[3] & 4) || b[12] != 'B' || b[13] != 'C') die("sort: a sorted run is damaged");
			blk_t k; k.at = o; k.bsize = (size_t)(b[16] | (size_t)b[17] << 8) + 1; k.xlen = b[10] | (size_t)b[11] << 8;
			if (o + k.bsize > raw.size() || k.bsize < 12 + k.xlen + 8) die("sort: a sorted run is damaged");
			memcpy(&k.isz, b + k.bsize - 4, 4); k.uo = uo; uo += k.isz; o += k.bsize;
			blk.push_back(k);
		}
		if (uo != u1 - u0) die("sort: a sorted run is damaged");
		fu_buf_t buf; if (!buf.heap((size_t)uo)) die("sort: out of memory");
		parallel_for((int)std::min<size_t>((size_t)std::max(1, threads), blk.size() / 4 + 1), blk.size(), [&](size_t a, size_t b, int) {
			z_stream zs; memset(&zs, 0, sizeof(zs));
			if (inflateInit2(&zs, -15) != Z_OK) die("sort: zlib");
			for (size_t k = a; k < b; ++k) {
				const blk_t &B = blk[k];
				if (!B.isz) continue;
				zs.next_in = (Bytef*)(raw.data() + B.at + 12 + B.xlen); zs.avail_in = (uInt)(B.bsize - 12 - B.xlen - 8); zs.next_out = buf.p + B.uo; zs.avail_out = B.isz;
				if (inflateReset(&zs) != Z_OK || inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out) die("sort: a sorted run does not inflate");
			}
			inflateEnd(&zs);
		});
		const size_t id = S.chunk.size();
		for (size_t g = g0; g < g1; ++g) if (R.useg[g + 1] > R.useg[g]) { piece_t P; P.chunk = id; P.run = r; P.g = g; P.a = R.useg[g] - u0; P.b = R.useg[g + 1] - u0; pieces.push_back(std::move(P)); }
		S.chunk.push_back(std::move(buf)); S.chunk_len.push_back((size_t)uo); S.bytes += uo;
	}
	std::atomic<int> bad(0);
	parallel_for((int)std::min<size_t>((size_t)std::max(1, threads), pieces.size()), pieces.size(), [&](size_t a, size_t b, int) {
		for (size_t k = a; k < b; ++k) {
			piece_t &P = pieces[k]; const uint8_t *p = S.chunk[P.chunk].p; uint64_t o = P.a;
			while (o + 4 <= P.b) { uint32_t bs; memcpy(&bs, p + o, 4); if (o + 4 + (uint64_t)bs > P.b || bs < 32) { bad = 1; break; } P.loc.push_back((uint64_t)P.chunk << 40 | o); P.key.push_back(bam_sort_key(p + o + 4)); o += 4 + (uint64_t)bs; }
			if (o != P.b) bad = 1;
		}
	});
	if (bad) die("sort: a sorted run is damaged");
	size_t n = 0; for (const piece_t &P : pieces) n += P.loc.size();
	if (n >= 0xfffffff0u) die("sort: a stretch of the merge holds too many records (raise -m)");
	S.loc.resize(n); S.key.resize(n);
	{ std::vector<size_t> at(pieces.size() + 1, 0); for (size_t k = 0; k < pieces.size(); ++k) at[k + 1] = at[k] + pieces[k].loc.size();
	  const bool with_ord = !runs.empty() && runs[0].ord_fd >= 0;   /* rank mode: the records' input ordinals, from the runs' side files */
	  if (with_ord) S.ord.resize(n);
	  parallel_for((int)std::min<size_t>((size_t)std::max(1, threads), pieces.size()), pieces.size(), [&](size_t a, size_t b, int) {
		for (size_t k = a; k < b; ++k) {
			const piece_t &P = pieces[k];
			if (P.loc.empty()) continue;
			memcpy(&S.loc[at[k]], P.loc.data(), 8 * P.loc.size()); memcpy(&S.key[at[k]], P.key.data(), 8 * P.key.size());
			if (with_ord) {
				const run_t &R = runs[P.run];
				if (R.rcnt[P.g + 1] - R.rcnt[P.g] != P.loc.size()) { bad = 1; continue; }
				uint8_t *d = (uint8_t*)&S.ord[at[k]]; const size_t want = 8 * P.loc.size();
				for (size_t got = 0; got < want; ) { const ssize_t q = pread(R.ord_fd, d + got, want - got, (off_t)(8 * R.rcnt[P.g] + got)); if (q < 0 && errno == EINTR) continue; if (q <= 0) { bad = 1; break; } got += (size_t)q; }
			}
		} }); }
	if (bad) die("sort: a sorted run's ordinals are damaged");
}

/* order of a stretch's records: by key, equal keys by input ordinal when there are ordinals (rank mode: runs of several ranks hold records of
 * interleaved batches), else by position in the store (stable sort: earlier run first) */
static void stretch_perm(const rec_store_t &S, std::vector<uint32_t> &perm)
{
	if (S.ord.empty()) { gpu_perm(S, perm); return; }
	const size_t n = S.key.size();
	std::vector<uint32_t> p1(n), p2(n); std::vector<uint64_t> k1(n);
	if (ssg_sort_u64_perm(S.ord.data(), (int64_t)n, p1.data())) die(std::string("sort: ") + ssg_last_error());
	for (size_t i = 0; i < n; ++i) k1[i] = S.key[p1[i]];
	if (ssg_sort_u64_perm(k1.data(), (int64_t)n, p2.data())) die(std::string("sort: ") + ssg_last_error());
	perm.resize(n);
	for (size_t i = 0; i < n; ++i) perm[i] = p1[p2[i]];
}

/* g_lo .. g_hi: the ranges this call writes (all of them, or this rank's share in rank mode); seg_ret: where the header ended (rank mode's parts) */
static void merge_runs(std::vector<run_t> &runs, size_t G, const bam_hdr_t &h, int fd, int level, int threads, uint64_t budget, const char *bai_path, size_t g_lo = 0, size_t g_hi = (size_t)-1, uint64_t *hdr_end_ret = 0, const char *ent_path = 0)
{
	if (g_hi == (size_t)-1) g_hi = G;
	std::vector<size_t> cutg(1, g_lo);
	{	const uint64_t target = std::max<uint64_t>(budget / 3, 1); uint64_t acc = 0;
		for (size_t g = g_lo; g < g_hi; ++g) {
			uint64_t sz = 0; for (const run_t &R : runs) sz += R.useg[g + 1] - R.useg[g];
			if (acc && acc + sz > target) { cutg.push_back(g); acc = 0; }
			acc += sz;
		}
		cutg.push_back(g_hi);
	}
	const size_t ns = cutg.size() - 1;
	chan_t<std::unique_ptr<rec_store_t> > ch(1);
	std::thread loader([&]() {
		for (size_t k = 0; k < ns; ++k) { std::unique_ptr<rec_store_t> S(new rec_store_t()); load_stretch(runs, cutg[k], cutg[k + 1], threads, *S); ch.push(std::move(S)); }
		ch.close();
	});
	seg_out_t seg; double t_wait = 0, t_perm = 0, t_write = 0;
	if (ent_path) { seg.ent_fd = open(ent_path, O_WRONLY | O_CREAT | O_TRUNC, 0644); if (seg.ent_fd < 0) die(std::string("sort: cannot write ") + ent_path); }
	for (size_t k = 0; k < ns; ++k) {
		std::unique_ptr<rec_store_t> S;
		{ const double t0 = wall(); if (!ch.pop(S)) die("sort: the merge lost a stretch"); t_wait += wall() - t0; }
		seg.first = k == 0; seg.last = k + 1 == ns;
		std::vector<uint32_t> perm;
		{ const double t0 = wall(); stretch_perm(*S, perm); t_perm += wall() - t0; }
		{ const double t0 = wall(); write_sorted(*S, perm, h, fd, level, threads, bai_path, 0, 0, &seg); t_write += wall() - t0; }
	}
	loader.join();
	if (seg.ent_fd >= 0) close(seg.ent_fd);
	if (hdr_end_ret) *hdr_end_ret = seg.hdr_end;
	if (dbg()) fprintf(stderr, "[sambamba] sort: merge: %zu stretches of the genome; waited %.2f s for the loader (read + inflate + index of the runs), device sort of the keys %.2f s, gather + deflate + write %.2f s\n", ns, t_wait, t_perm, t_write);
}

/* rank mode, after a rank's part is written (by the merge of all ranks' runs, or from the blocks of the collective exchange): the markers the other ranks and
 * bin/speedseq-ranks wait for, and this rank's stretch placed in the joined file */
static bool place_part(const std::string &outp, const std::string &rdv, int rank, int world, uint64_t hdr_end)
{
		{ char t[64]; snprintf(t, sizeof(t), "%llu\n", (unsigned long long)hdr_end); if (!rk_file_put(outp + ".ssg_part", t, strlen(t))) die("sort: cannot write " + outp + ".ssg_part"); }
		/* the runs may go when every rank has read what it needed of them; the marker says how long this rank's part is and where its header blocks end */
		struct stat psb; if (stat(outp.c_str(), &psb) != 0) die("sort: cannot stat " + outp);
		{ char t[96]; snprintf(t, sizeof(t), "%llu %llu\n", (unsigned long long)psb.st_size, (unsigned long long)hdr_end); if (!rk_file_put(rdv + "/merged." + std::to_string(rank), t, strlen(t))) die("sort: cannot write into " + rdv); }
		for (int r = 0; r < world; ++r) if (!rk_file_wait(rdv + "/merged." + std::to_string(r))) die("sort: rank " + std::to_string(r) + " did not finish its merge");
		/* Every rank places its own stretch in the joined file (PREFIX.bam for a part named PREFIX.rank<r>.bam) at the offset the parts before it leave -- the
		 * stretches' lengths are only known now, so this is a copy, but N of them side by side instead of one by the launcher after the last rank has ended
		 * (the one term of the ranks' wall time that grew with the input and did not divide by N).  Part 0 goes with its header blocks, the others without; the
		 * last rank adds the end-of-file block; bin/speedseq-ranks finds joined.<r> of every rank and only indexes.  SSG_RANKS_JOIN=0 leaves the copy to it. */
		const std::string tail = ".rank" + std::to_string(rank) + ".bam";
		if (!(getenv("SSG_RANKS_JOIN") && !strcmp(getenv("SSG_RANKS_JOIN"), "0")) && outp.size() > tail.size() && outp.compare(outp.size() - tail.size(), tail.size(), tail) == 0) {
			const double tj = wall();
			uint64_t at = 0, my_from = 0, my_len = 0; bool ok = true;
			for (int r = 0; r < world && ok; ++r) {
				std::vector<uint8_t> b; unsigned long long sz = 0, he = 0;
				if (!rk_file_get(rdv + "/merged." + std::to_string(r), b)) { ok = false; break; }
				b.push_back(0);
				if (sscanf((const char*)b.data(), "%llu %llu", &sz, &he) != 2 || sz < 28 + he) { ok = false; break; }
				const uint64_t from = r ? he : 0, len = sz - 28 - from;
				if (r == rank) { my_from = from; my_len = len; break; }
				at += len;
			}
			const std::string joined = outp.substr(0, outp.size() - tail.size()) + ".bam";
			int jfd = ok ? open(joined.c_str(), O_WRONLY | O_CREAT, 0644) : -1, pfd = ok ? open(outp.c_str(), O_RDONLY) : -1;
			if (jfd >= 0 && pfd >= 0) {
				std::vector<uint8_t> buf;
				uint64_t done_b = 0;
				while (done_b < my_len && ok) {
					off64_t oi = (off64_t)(my_from + done_b), oo = (off64_t)(at + done_b);
					ssize_t k = copy_file_range(pfd, &oi, jfd, &oo, (size_t)std::min<uint64_t>(my_len - done_b, (uint64_t)1 << 30), 0);
					if (k <= 0) {   /* a file system that cannot: through a buffer */
						if (buf.empty()) buf.resize((size_t)8 << 20);
						k = pread(pfd, buf.data(), (size_t)std::min<uint64_t>(my_len - done_b, buf.size()), (off_t)(my_from + done_b));
						if (k <= 0) { ok = false; break; }
						for (ssize_t w = 0; w < k; ) { const ssize_t x = pwrite(jfd, buf.data() + w, (size_t)(k - w), (off_t)(at + done_b + (uint64_t)w)); if (x <= 0) { ok = false; break; } w += x; }
					}
					done_b += (uint64_t)k;
				}
				if (ok && rank == world - 1 && pwrite(jfd, BGZF_EOF, 28, (off_t)(at + my_len)) != 28) ok = false;
			} else ok = false;
			if (jfd >= 0 && close(jfd) != 0) ok = false;
			if (pfd >= 0) close(pfd);
			if (!ok) die("sort: cannot place this rank's stretch in " + joined);
			if (!rk_file_put(rdv + "/joined." + std::to_string(rank), "", 0)) die("sort: cannot write into " + rdv);
			if (dbg()) fprintf(stderr, "[sambamba] sort: rank %d placed its stretch (%.2f GB) at byte %llu of %s in %.2f s\n", rank, (double)my_len / 1e9, (unsigned long long)at, joined.c_str(), wall() - tj);
		}
	return true;
}

static int cmd_sort(int argc, char **argv)
{
	int threads = hw_threads(), level = -1; double mem_gb = 2; std::string tmpdir = ".", outp; const char *in = 0;
	for (int i = 0; i < argc; ++i) {
		const char *a = argv[i];
		if (!strcmp(a, "-t") && i + 1 < argc) threads = atoi(argv[++i]);
		else if (!strcmp(a, "-m") && i + 1 < argc) { const char *v = argv[++i]; char *e; mem_gb = strtod(v, &e); if (*e == 'M' || *e == 'm') mem_gb /= 1024; else if (*e == 'K' || *e == 'k') mem_gb /= 1048576; else if (!*e) mem_gb /= 1073741824.0; }
		else if (!strncmp(a, "--tmpdir=", 9)) tmpdir = a + 9;
		else if (!strcmp(a, "--tmpdir") && i + 1 < argc) tmpdir = argv[++i];
		else if (!strcmp(a, "-o") && i + 1 < argc) outp = argv[++i];
		else if (!strcmp(a, "-l") && i + 1 < argc) level = atoi(argv[++i]);
		else if (a[0] == '-' && a[1] && strcmp(a, "-")) die(std::string("sort: unsupported option ") + a);
		else in = a;
	}
	if (!in || outp.empty()) die("usage: sambamba sort [-t N] [-m XG] [--tmpdir=DIR] -o out.bam <in.bam>");
	if (threads < 1) threads = 1;
	const double t_start = wall();
	{ const char *e = getenv("SSG_BAM_LEVEL"); if (level < 0 && e && *e) level = atoi(e); }   /* deflate level of the sorted file when -l is not given (default: zlib's 6, as sambamba's) */
	/* compression is CPU work the reference's `-t` undersizes on a host with hundreds of cores next to an MI355X: the pool may use more (SSG_SORT_THREADS) */
	int pool = threads; { const char *e = getenv("SSG_SORT_THREADS"); if (e && atoi(e) > 0) pool = atoi(e); }
	pool = ssg_pool_threads(pool);
	const int fd = open_in(in);
	char first[8]; size_t n_first = 0;
	while (n_first < 8) { ssize_t r = read(fd, first + n_first, 8 - n_first); if (r < 0) { if (errno == EINTR) continue; die("sort: read error"); } if (r == 0) break; n_first += (size_t)r; }
	const bool fused = n_first == 8 && !memcmp(first, FU_MAGIC, 8);
	/* Rank mode (ranks.h): this sort holds the records of its rank's batches; all ranks' sorts exchange sorted runs through SSG_RDV and each
	 * writes the stretch of the genome it owns, as a BAM of its own that bin/speedseq-ranks joins to the others.  Only the main stream (frames)
	 * is exchanged: the side streams were brought to rank 0 by the samblasters already. */
	const int world = fused ? rk_world() : 1, rank = rk_rank();
	if (world > 1 && !rk_check("sambamba")) return 1;
	const std::string rdv = rk_dir(), rdata = rk_data_dir();
	/* the transport of the sorts' exchange (xchg.h: RCCL, sockets, or none = the files of rounds 4-5) comes up while the input streams in */
	xchg_t *X = 0; std::thread t_xchg;
	if (world > 1) t_xchg = std::thread([&]() { X = xchg_open(rank, world, rdv); });
	uint64_t budget = (uint64_t)(std::max(mem_gb, 0.25) * 0.6 * 1073741824.0);   /* record bytes per in-memory run; the rest is keys, locations, output blocks */
	{ const char *e = getenv("SSG_SORT_CHUNK_BYTES"); if (e && atoll(e) > 0) budget = (uint64_t)atoll(e); }   /* the tests force the spill-and-merge path */
	rec_store_t S; std::vector<std::string> spills; bam_hdr_t h;
	/* the device is first needed when the last record has arrived: bring the runtime up now, next to the input, not then */
	std::thread warm([]() { const uint64_t k[2] = { 1, 0 }; uint32_t pm[2]; (void)ssg_sort_u64_perm(k, 2, pm); });
	struct joiner_t { std::thread &t; ~joiner_t() { if (t.joinable()) t.join(); } } warm_join = { warm };
	if (mkdir(tmpdir.c_str(), 0777) != 0 && errno != EEXIST) die("sort: cannot create " + tmpdir + ": " + strerror(errno));
	std::vector<run_t> runs; std::vector<uint64_t> range_lo;
	/* One run at a time; in the fused single-pipeline sort a run is written by a thread of its own while the input goes on into a second store (each of half the
	 * budget): until round 6 the input -- and with it samblaster and `bwa mem` -- stood still for as long as a run took to write. */
	rec_store_t S_bg; std::thread t_spill;
	auto spill_wait = [&]() { if (t_spill.joinable()) t_spill.join(); };
	auto spill_store = [&](rec_store_t &S) {
		std::vector<uint32_t> perm; gpu_perm(S, perm);
		if (runs.empty()) {   /* the ranges of the genome, fixed now: about 4 MB of a run each, so that one range of all runs is a small merge */
			size_t G = world > 1 ? 1024 : (size_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, S.bytes >> 22));   /* rank mode: the same ranges on every rank */
			{ const char *e = getenv("SSG_SORT_RANGES"); if (e && atol(e) > 0) G = (size_t)atol(e); }
			make_ranges(h, G, range_lo);
		}
		const size_t G = range_lo.size(), n = perm.size();
		std::vector<size_t> at(G + 1, n);
		for (size_t g = 0; g < G; ++g) {   /* first record of the sorted order whose key reaches the range */
			size_t a = 0, b = n;
			while (a < b) { const size_t m = (a + b) >> 1; if (S.key[perm[m]] < range_lo[g]) a = m + 1; else b = m; }
			at[g] = a;
		}
		char nm[64]; snprintf(nm, sizeof(nm), "/ssg_sort_%d_%04zu.run", (int)getpid(), runs.size());
		if (world > 1) snprintf(nm, sizeof(nm), "/run.%d.%zu.run", rank, runs.size());
		run_t R; R.path = (world > 1 ? rdata : tmpdir) + nm;
		R.fd = open(R.path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644); if (R.fd < 0) die("sort: cannot write " + R.path);
		if (world > 1) {   /* equal keys keep their input order over all ranks: by ordinal within the run (the exchange sorts by it across runs) */
			std::vector<uint32_t> p1(n), p2(n); std::vector<uint64_t> k1(n);
			if (n && ssg_sort_u64_perm(S.ord.data(), (int64_t)n, p1.data())) die(std::string("sort: ") + ssg_last_error());
			for (size_t i = 0; i < n; ++i) k1[i] = S.key[p1[i]];
			if (n && ssg_sort_u64_perm(k1.data(), (int64_t)n, p2.data())) die(std::string("sort: ") + ssg_last_error());
			for (size_t i = 0; i < n; ++i) perm[i] = p1[p2[i]];
		}
		write_sorted(S, perm, h, R.fd, 1, pool, 0, &at, &R.seg, 0, &R.useg);
		if (world > 1) {   /* what the other ranks need of this run: segment offsets, record counts, ordinals in run order */
			std::vector<uint64_t> idx; idx.push_back((uint64_t)G);
			idx.insert(idx.end(), R.seg.begin(), R.seg.end()); idx.insert(idx.end(), R.useg.begin(), R.useg.end());
			for (size_t g = 0; g <= G; ++g) idx.push_back((uint64_t)at[g]);
			std::vector<uint64_t> od(n); for (size_t i = 0; i < n; ++i) od[i] = S.ord[perm[i]];
			if (!rk_file_put(R.path + ".ord", od.data(), 8 * n) || !rk_file_put(R.path + ".idx", idx.data(), 8 * idx.size())) die("sort: cannot write into " + rdata);
		}
		runs.push_back(R); spills.push_back(R.path); S.clear();
	};
	auto spill = [&]() { spill_wait(); spill_store(S); };
	const bool spill_bg = fused && world == 1 && !(getenv("SSG_SORT_SPILL_BG") && atoi(getenv("SSG_SORT_SPILL_BG")) == 0);
	auto spill_async = [&]() { spill_wait(); std::swap(S, S_bg); t_spill = std::thread([&]() { spill_store(S_bg); }); };
	struct spill_join_t { std::thread &t; ~spill_join_t() { if (t.joinable()) t.join(); } } spill_join = { t_spill };
	const uint64_t budget_in = spill_bg ? std::max<uint64_t>(budget / 2, 1) : budget;
	if (fused) {
		/* frames straight from samblaster (fused.h): a reader thread takes them off the pipe, this thread indexes the records (keys +
		 * locations) of each while the next arrives -- the sort's input work overlaps the alignment upstream */
		struct frame_t { fu_frame_t fh; fu_buf_t p; };
		chan_t<std::unique_ptr<frame_t> > ch(2); std::atomic<int> rd_fail(0);
		std::thread reader([&]() {
			for (;;) {
				std::unique_ptr<frame_t> F(new frame_t());
				if (!fu_read_frame(fd, F->fh, F->p)) { rd_fail = 1; fu_discard_rest(fd); break; }
				const bool end = F->fh.type == FU_END;
				ch.push(std::move(F));
				if (end) break;
			}
			ch.close();
		});
		std::unique_ptr<frame_t> F; bool ended = false; double t_wait = 0, t_index = 0; const char *bad = 0; uint64_t n_main = 0;   /* MAIN frames so far: frame k of rank r holds batch r + k * world */
		/* the page-locked staging blocks of the writer's producers (write_sorted: three producers, two slots and an output block each, 134 MB apiece) are made now, next
		 * to the input, and wait in the library's pool: page-locking 1.2 GB when the last record has arrived was a fifth of a second of the sort's tail.  Started with
		 * the first record frame -- by then `bwa mem' has its index on the device (the driver serialises page-locked allocations: see bwa_main.cpp). */
		std::thread prewarm; bool prewarmed = level == 0 || (getenv("SSG_SORT_PREWARM") && atoi(getenv("SSG_SORT_PREWARM")) == 0);
		struct prewarm_join_t { std::thread &t; ~prewarm_join_t() { if (t.joinable()) t.join(); } } prewarm_join = { prewarm };
		for (;;) {
			{ const double t0 = wall(); const bool got = ch.pop(F); t_wait += wall() - t0; if (!got) break; }
			if (F->fh.type == FU_END) { ended = true; break; }
			if (F->fh.type == FU_HEADER) { h.text.assign((const char*)F->p.p, (size_t)F->fh.len); hdr_from_text(h); change_so(h.text, "coordinate"); continue; }
			if (F->fh.type != FU_MAIN) { bad = "sort: unexpected frame in the fused stream"; break; }
			if (!F->fh.len) { ++n_main; continue; }
			if (!prewarmed) {
				prewarmed = true;
				prewarm = std::thread([]() {
					if (ssg_device_count() < 1 || !strcmp(ssg_backend(), "emu")) return;
					std::vector<void*> b;
					for (int k = 0; k < 9; ++k) b.push_back(ssg_host_alloc((size_t)2048 * (BGZF_MAX_PAYLOAD + 5) + 64));
					for (void *p : b) ssg_host_free(p);
				});
			}
			{ const double t0 = wall(); if (!S.add_chunk(std::move(F->p), (size_t)F->fh.len, world > 1 ? ((uint64_t)rank + n_main * (uint64_t)world) << 28 : ~(uint64_t)0)) { bad = "sort: malformed record frame"; break; } t_index += wall() - t0; }
			++n_main;
			if (S.bytes >= budget_in || S.key.size() >= 0xfffffff0u) { if (spill_bg) spill_async(); else spill(); }
		}
		spill_wait();
		if (dbg()) fprintf(stderr, "[sambamba] sort: input thread waited %.2f s for frames, indexed records for %.2f s\n", t_wait, t_index);
		{ std::unique_ptr<frame_t> drop; while (ch.pop(drop)) {} }
		reader.join();                                           /* the rest of the stream was taken in (and its segments released) even after an error */
		if (bad) die(bad);
		if (rd_fail || !ended) die("sort: the fused stream ended early");
		if (h.names.empty() && h.text.empty()) change_so(h.text, "coordinate");
	} else {
		bgzf_in_t bi(fd, threads);
		bi.raw.assign((const uint8_t*)first, (const uint8_t*)first + n_first);
		if (!hdr_read(bi, h)) die("sort: not a BAM file");
		change_so(h.text, "coordinate");
		const size_t CH = (size_t)std::min<uint64_t>((uint64_t)64 << 20, std::max<uint64_t>(budget / 4, 65536));   /* the budget is checked once per chunk */
		fu_buf_t cur; size_t cap = CH, len = 0;
		if (!cur.heap(CH)) die("sort: out of memory");
		auto flush = [&]() { if (!len) return; if (!S.add_chunk(std::move(cur), len)) die("sort: malformed BAM record"); cap = CH; len = 0; if (!cur.heap(CH)) die("sort: out of memory"); if (S.bytes >= budget || S.key.size() >= 0xfffffff0u) spill(); };
		for (;;) {
			uint32_t bs;
			if (bi.get(&bs, 4) != 4) break;
			if (len + 4 + (size_t)bs > cap) {
				flush();
				if (4 + (size_t)bs > cap) { cap = 4 + (size_t)bs; if (!cur.heap(cap)) die("sort: out of memory"); }
			}
			memcpy(cur.p + len, &bs, 4);
			if (bi.get(cur.p + len + 4, bs) != bs) die("sort: truncated BAM");
			len += 4 + (size_t)bs;
		}
		flush();
	}
	const double t_in = wall();
	ssg_stamp("sambamba_sort", "input_done");
	bool bai_note = false;
	int ofd = open(outp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); if (ofd < 0) die("sort: cannot write " + outp);
	if (world > 1) {
		t_xchg.join();
		if (X == (xchg_t*)-1) die("sort: the ranks' exchange transport failed to come up");
		if (X) {
			/* The collective form of the exchange (SURVEY.md 8e coupling 3): the records of this rank in (key, ordinal) order, cut at the stretches of the genome the ranks own (the
			 * same 1024 ranges as the file form, so the same records land on the same rank); sizes first, then one all-to-all of the blocks (ordinals + record bytes); the
			 * receiver orders what it got by (key, ordinal) on its device and writes its part.  Whether everybody can -- no rank has spilled, every rank's stretch fits its
			 * budget -- is itself agreed by an exchange; otherwise all ranks go on through the files. */
			const double t_x = wall();
			const size_t G = 1024; make_ranges(h, G, range_lo);
			std::vector<uint32_t> perm; stretch_perm(S, perm);
			const size_t n = perm.size();
			std::vector<size_t> cutp((size_t)world + 1, n);
			for (int d = 0; d < world; ++d) {
				const uint64_t lo_key = range_lo[G * (size_t)d / (size_t)world];
				size_t a = 0, b = n;
				while (a < b) { const size_t m = (a + b) >> 1; if (S.key[perm[m]] < lo_key) a = m + 1; else b = m; }
				cutp[(size_t)d] = a;
			}
			cutp[0] = 0;
			std::vector<uint64_t> cs((size_t)world * 3), cr((size_t)world * 3);
			for (int d = 0; d < world; ++d) {
				uint64_t bytes = 0;
				for (size_t i = cutp[(size_t)d]; i < cutp[(size_t)d + 1]; ++i) { uint32_t bs; memcpy(&bs, S.rec(perm[i]), 4); bytes += 4 + (uint64_t)bs; }
				cs[(size_t)d * 3] = cutp[(size_t)d + 1] - cutp[(size_t)d]; cs[(size_t)d * 3 + 1] = bytes; cs[(size_t)d * 3 + 2] = runs.empty() ? 1 : 0;
			}
			if (!X->alltoall_u64(cs.data(), cr.data(), 3)) die("sort: the ranks' exchange failed");
			uint64_t in_bytes = 0; bool all_ok = true;
			for (int q = 0; q < world; ++q) { in_bytes += cr[(size_t)q * 3 + 1]; if (!cr[(size_t)q * 3 + 2]) all_ok = false; }
			std::vector<uint64_t> f1((size_t)world, (all_ok && in_bytes <= budget * 2) ? 1 : 0), f2((size_t)world, 0);
			if (!X->alltoall_u64(f1.data(), f2.data(), 1)) die("sort: the ranks' exchange failed");
			for (int q = 0; q < world; ++q) if (!f2[(size_t)q]) all_ok = false;
			if (!all_ok) { if (rank == 0 && dbg()) fprintf(stderr, "[sambamba] sort: a rank's share does not fit its memory (or its input did not): the ranks exchange through files\n"); delete X; X = 0; }
			else {
				std::vector<std::vector<uint8_t> > sb((size_t)world), rb((size_t)world);
				std::vector<const void*> sp((size_t)world); std::vector<void*> rp((size_t)world); std::vector<uint64_t> sn((size_t)world), rn((size_t)world);
				parallel_for(std::min(pool, world), (size_t)world, [&](size_t a, size_t b, int) {
					for (size_t d = a; d < b; ++d) {
						const size_t nd = cutp[d + 1] - cutp[d];
						sb[d].resize(8 * nd + (size_t)cs[d * 3 + 1]);
						uint8_t *w = sb[d].data() + 8 * nd;
						for (size_t i = 0; i < nd; ++i) {
							const uint32_t id = perm[cutp[d] + i]; memcpy(sb[d].data() + 8 * i, &S.ord[id], 8);
							const uint8_t *r = S.rec(id); uint32_t bs; memcpy(&bs, r, 4); memcpy(w, r, 4 + (size_t)bs); w += 4 + (size_t)bs;
						}
					}
				});
				for (int q = 0; q < world; ++q) { rb[(size_t)q].resize(8 * (size_t)cr[(size_t)q * 3] + (size_t)cr[(size_t)q * 3 + 1]); sp[(size_t)q] = sb[(size_t)q].data(); rp[(size_t)q] = rb[(size_t)q].data(); sn[(size_t)q] = sb[(size_t)q].size(); rn[(size_t)q] = rb[(size_t)q].size(); }
				S.clear();
				if (!X->alltoallv(sp.data(), sn.data(), rp.data(), rn.data())) die("sort: the ranks' exchange failed");
				for (auto &v : sb) { std::vector<uint8_t>().swap(v); }
				rec_store_t R;
				for (int q = 0; q < world; ++q) {
					const size_t nq = (size_t)cr[(size_t)q * 3], len = (size_t)cr[(size_t)q * 3 + 1];
					if (!nq) continue;
					fu_buf_t c; if (!c.heap(len)) die("sort: out of memory");
					memcpy(c.p, rb[(size_t)q].data() + 8 * nq, len);
					std::vector<uint64_t> od(nq); memcpy(od.data(), rb[(size_t)q].data(), 8 * nq);
					std::vector<uint8_t>().swap(rb[(size_t)q]);
					if (!add_chunk_ords(R, std::move(c), len, od.data(), nq)) die("sort: a block of the ranks' exchange is not whole records");
				}
				std::vector<uint32_t> pr; stretch_perm(R, pr);
				seg_out_t seg; seg.first = true; seg.last = true;
				seg.ent_fd = open((outp + ".ssg_ent").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); if (seg.ent_fd < 0) die("sort: cannot write " + outp + ".ssg_ent");
				write_sorted(R, pr, h, ofd, level, pool, 0, 0, 0, &seg);
				close(seg.ent_fd); close(ofd);
				const uint64_t hdr_end = seg.hdr_end;
				if (dbg()) fprintf(stderr, "[sambamba] sort: rank %d of %d: %zu records of its own in, %zu records of its stretch out, exchanged over %s: %.2f s\n", rank, world, n, pr.size(), X->name(), wall() - t_x);
				delete X; X = 0;
				if (!place_part(outp, rdv, rank, world, hdr_end)) die("sort: cannot place this rank's stretch");
				return 0;
			}
		}
		/* every record of this rank goes into exchange runs (the last one now); then all ranks' runs are opened and this rank merges its share of the ranges */
		if (!S.key.empty() || runs.empty()) spill();
		const double t_x = wall();
		{ const uint64_t nr = runs.size(); if (!rk_file_put(rdv + "/sorted." + std::to_string(rank), &nr, 8)) die("sort: cannot write into " + rdv); }
		std::vector<run_t> all; size_t G = range_lo.size();
		for (int r = 0; r < world; ++r) {
			const std::string dn = rdv + "/sorted." + std::to_string(r); std::vector<uint8_t> b;
			if (!rk_file_wait(dn) || !rk_file_get(dn, b) || b.size() != 8) die("sort: rank " + std::to_string(r) + " did not deliver its runs");
			uint64_t nr; memcpy(&nr, b.data(), 8);
			for (uint64_t k = 0; k < nr; ++k) {
				run_t R; R.path = rdata + "/run." + std::to_string(r) + "." + std::to_string(k) + ".run";
				std::vector<uint8_t> ib;
				if (!rk_file_get(R.path + ".idx", ib) || ib.size() < 8) die("sort: cannot read " + R.path + ".idx");
				uint64_t g; memcpy(&g, ib.data(), 8);
				if (g != G || ib.size() != 8 * (1 + 3 * (g + 1))) die("sort: the ranks disagree about the ranges of the genome (different headers?)");
				const uint64_t *q = (const uint64_t*)(ib.data() + 8);
				R.seg.assign(q, q + g + 1); R.useg.assign(q + g + 1, q + 2 * (g + 1)); R.rcnt.assign(q + 2 * (g + 1), q + 3 * (g + 1));
				R.fd = open(R.path.c_str(), O_RDONLY); R.ord_fd = open((R.path + ".ord").c_str(), O_RDONLY);
				if (R.fd < 0 || R.ord_fd < 0) die("sort: cannot open " + R.path);
				all.push_back(R);
			}
		}
		const size_t g_lo = G * (size_t)rank / (size_t)world, g_hi = G * ((size_t)rank + 1) / (size_t)world;
		uint64_t hdr_end = 0;
		merge_runs(all, G, h, ofd, level, pool, budget, 0, g_lo, g_hi, &hdr_end, (outp + ".ssg_ent").c_str());
		close(ofd);
		if (!place_part(outp, rdv, rank, world, hdr_end)) die("sort: cannot place this rank's stretch");
		for (run_t &R : all) { close(R.fd); close(R.ord_fd); }
		for (run_t &R : runs) { close(R.fd); unlink(R.path.c_str()); unlink((R.path + ".ord").c_str()); unlink((R.path + ".idx").c_str()); }
		if (dbg()) fprintf(stderr, "[sambamba] sort: rank %d of %d: input %.2f s (from start), %zu run(s) of its own, ranges %zu .. %zu of %zu from %zu runs of all ranks: %.2f s\n", rank, world, t_in - t_start, runs.size(), g_lo, g_hi, G, all.size(), wall() - t_x);
		return 0;
	}
	if (spills.empty()) {
		std::vector<uint32_t> perm; gpu_perm(S, perm);
		const double t_perm = wall();
		const std::string bai = outp + ".bai";
		write_sorted(S, perm, h, ofd, level, pool, getenv("SSG_SORT_NO_BAI") ? 0 : bai.c_str());
		bai_note = !getenv("SSG_SORT_NO_BAI");
		if (dbg()) fprintf(stderr, "[sambamba] sort: %zu records, %.2f GB: input %.2f s (from start), device sort of the keys %.2f s, gather + deflate (level %d, %d threads) + write %.2f s\n",
		                   S.key.size(), (double)S.bytes / 1e9, t_in - t_start, t_perm - t_in, level < 0 ? 6 : level, pool, wall() - t_perm);
	}
	else {
		if (!S.key.empty()) spill();
		const double t_m = wall();
		const std::string bai = outp + ".bai";
		merge_runs(runs, range_lo.size(), h, ofd, level, pool, budget, getenv("SSG_SORT_NO_BAI") ? 0 : bai.c_str());
		bai_note = !getenv("SSG_SORT_NO_BAI");
		if (dbg()) fprintf(stderr, "[sambamba] sort: input %.2f s (from start), %zu sorted runs merged in %zu ranges of the genome by %d threads: %.2f s\n", t_in - t_start, runs.size(), range_lo.size(), pool, wall() - t_m);
		for (run_t &R : runs) { close(R.fd); unlink(R.path.c_str()); }
	}
	close(ofd);
	if (bai_note) bai_note_write(outp);
	ssg_stamp("sambamba_sort", "written");
	ssg_worker_done(0);         /* (the record store's chunks are gigabytes of mapped frames: the script does not wait for them to be unmapped) */
	return ssg_fast_exit(0);
}

/* ---------------- index ---------------- */
/* `sambamba flagstat <in.bam>` (upstream sambamba has the command; the reference does not call it): the counts of samtools flagstat that
 * the soak's invariants need, in its line format, from the fixed fields only -- blocks inflated by the pool, one pass of a 100 GB file
 * in the time the reference's samtools needs for a few GB -- plus one line samtools does not print: whether the file is in coordinate
 * order (bam1_lt's key, unmapped reads last). */
static int cmd_flagstat(int argc, char **argv)
{
	const char *in = 0; int threads = hw_threads();
	for (int i = 0; i < argc; ++i) { if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]); else if (argv[i][0] != '-') { if (!in) in = argv[i]; } }
	if (!in) die("usage: sambamba flagstat [-t N] <in.bam>");
	const int fd = open_in(in);
	bgzf_in_t bi(fd, threads); bam_hdr_t h;
	if (!hdr_read(bi, h)) die("flagstat: not a BAM file");
	uint64_t n = 0, sec = 0, sup = 0, dup = 0, mapped = 0, paired = 0, r1 = 0, r2 = 0, proper = 0, both = 0, single = 0, prim = 0, prim_dup = 0, descents = 0, prev = 0;
	uint8_t rec[32];
	for (;;) {
		uint32_t bs;
		if (bi.get(&bs, 4) != 4) break;
		if (bs < 32 || bi.get(rec, 32) != 32) die("flagstat: malformed BAM record");
		if (bi.skip(bs - 32) != bs - 32) die("flagstat: truncated BAM");
		bam_core_t c; memcpy(&c, rec, 32);
		const uint32_t f = c.flag_nc >> 16;
		++n;
		if (f & 0x100) ++sec; else if (f & 0x800) ++sup;
		if (f & 0x400) ++dup;
		if (!(f & 4)) ++mapped;
		if (!(f & 0x900)) { ++prim; if (f & 0x400) ++prim_dup; }
		if ((f & 1) && !(f & 0x900)) {
			++paired; if (f & 0x40) ++r1; if (f & 0x80) ++r2;
			if ((f & 2) && !(f & 4)) ++proper;
			if (!(f & 4) && !(f & 8)) ++both;
			if (!(f & 4) && (f & 8)) ++single;
		}
		const uint64_t key = (uint64_t)(uint32_t)c.tid << 32 | (uint32_t)((c.pos + 1) << 1) | ((f & 0x10) ? 1u : 0u);   /* tid -1 sorts last */
		if (n > 1 && key < prev) ++descents;
		prev = key;
	}
	close(fd);
	printf("%llu + 0 in total (QC-passed reads + QC-failed reads)\n%llu + 0 secondary\n%llu + 0 supplementary\n%llu + 0 duplicates\n%llu + 0 mapped\n%llu + 0 paired in sequencing\n%llu + 0 read1\n%llu + 0 read2\n"
	       "%llu + 0 properly paired\n%llu + 0 with itself and mate mapped\n%llu + 0 singletons\n%llu + 0 primary\n%llu + 0 primary duplicates\n%llu descents of the coordinate key (0 = sorted)\n",
	       (unsigned long long)n, (unsigned long long)sec, (unsigned long long)sup, (unsigned long long)dup, (unsigned long long)mapped, (unsigned long long)paired, (unsigned long long)r1, (unsigned long long)r2,
	       (unsigned long long)proper, (unsigned long long)both, (unsigned long long)single, (unsigned long long)prim, (unsigned long long)prim_dup, (unsigned long long)descents);
	return 0;
}

static int cmd_index(int argc, char **argv)
{
	const char *in = 0; int threads = hw_threads();
	for (int i = 0; i < argc; ++i) { if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]); else if (argv[i][0] != '-') { if (!in) in = argv[i]; } }
	if (!in) die("usage: sambamba index <in.bam>");
	if (rk_world() > 1) {   /* rank mode: a rank's stretch of the sorted file is not what gets indexed -- bin/speedseq-ranks indexes the joined file */
		struct stat sb;
		if (stat((std::string(in) + ".ssg_part").c_str(), &sb) == 0) { if (dbg()) fprintf(stderr, "[sambamba] index: %s is one rank's part of a sorted file: left to the join\n", in); return 0; }
	}
	{	/* the sort of this repository left the index of exactly this file next to it: nothing to do (the note goes, the pair stays) */
		const std::string note_p = std::string(in) + ".bai.ssg";
		FILE *f = fopen(note_p.c_str(), "r");
		if (f) {
			char have[128] = ""; if (!fgets(have, sizeof(have), f)) have[0] = 0;
			fclose(f); unlink(note_p.c_str());
			std::string now;
			if (bai_note_make(in, now) && now == have) { if (dbg()) fprintf(stderr, "[sambamba] index: %s.bai written by the sort is current\n", in); return 0; }
		}
	}
	const int fd = open_in(in);
	bgzf_in_t bi(fd, threads); bam_hdr_t h;
	if (!hdr_read(bi, h)) die("index: not a BAM file");
	bai_t idx((int)h.names.size(), bi.tell());
	std::vector<uint8_t> rec;
	for (;;) {   /* only the fixed fields, the name and the CIGAR of a record are looked at: the rest (sequence, qualities, tags) is skipped in place */
		uint32_t bs;
		if (bi.get(&bs, 4) != 4) break;
		if (bs < 32) die("index: malformed BAM record");
		rec.resize(32);
		if (bi.get(rec.data(), 32) != 32) die("index: truncated BAM");
		bam_core_t c; memcpy(&c, rec.data(), 32);
		const size_t var = (size_t)(c.bin_mq_nl & 0xff) + 4 * (size_t)(c.flag_nc & 0xffff);
		if (32 + var > bs) die("index: malformed BAM record");
		rec.resize(32 + var);
		if (bi.get(rec.data() + 32, var) != var || bi.skip(bs - 32 - var) != bs - 32 - var) die("index: truncated BAM");
		if (idx.push(c.tid, c.pos, bam_endpos(rec.data()), bi.tell(), !((c.flag_nc >> 16) & 4)) < 0) die("index: the file is not coordinate-sorted");
	}
	idx.finish(bi.tell());
	idx.save((std::string(in) + ".bai").c_str());
	return 0;
}

/* index of a file joined from the parts that the ranks of bin/speedseq-ranks wrote (ranks.h): every rank left, for each record of its part, the
 * fields an index entry is made of and the record's virtual offset in the part; `shift` moves a part's offsets to where its blocks lie in the
 * joined file.  The same bai_t calls in the same order as cmd_index makes from the file itself -- without inflating it again. */
static int cmd_index_parts(int argc, char **argv)
{
	if (argc < 3 || (argc - 1) % 2) die("usage: sambamba index-parts <joined.bam> <part.ssg_ent> <shift> [...]");
	const char *in = argv[0];
	const int fd = open_in(in);
	bgzf_in_t bi(fd, 2); bam_hdr_t h;
	if (!hdr_read(bi, h)) die("index-parts: not a BAM file");
	const uint64_t first = bi.tell();
	struct stat sb; if (fstat(fd, &sb) != 0) die("index-parts: cannot stat the file");
	const uint64_t eof_v = (uint64_t)sb.st_size << 16;
	bai_t idx((int)h.names.size(), first);
	bool have = false; int32_t e_tid = 0, e_pos = 0, e_end = 0; uint32_t e_m = 0;
	std::vector<uint8_t> buf((size_t)24 << 16);
	for (int a = 1; a + 1 < argc; a += 2) {
		const long long shift = atoll(argv[a + 1]);
		FILE *f = fopen(argv[a], "rb"); if (!f) die(std::string("index-parts: cannot read ") + argv[a]);
		for (;;) {
			const size_t k = fread(buf.data(), 24, buf.size() / 24, f);
			if (!k) break;
			for (size_t i = 0; i < k; ++i) {
				const uint8_t *d = buf.data() + 24 * i; uint64_t v; memcpy(&v, d + 16, 8);
				v = (uint64_t)((long long)(v >> 16) + shift) << 16 | (v & 0xffff);
				if (have && idx.push(e_tid, e_pos, e_end, v, e_m != 0) < 0) die("index-parts: the parts are not in coordinate order");
				memcpy(&e_tid, d, 4); memcpy(&e_pos, d + 4, 4); memcpy(&e_end, d + 8, 4); memcpy(&e_m, d + 12, 4); have = true;
			}
		}
		fclose(f);
	}
	if (have && idx.push(e_tid, e_pos, e_end, eof_v, e_m != 0) < 0) die("index-parts: the parts are not in coordinate order");
	idx.finish(eof_v);
	idx.save((std::string(in) + ".bai").c_str());
	close(fd);
	return 0;
}

/* ---------------- merge ---------------- */
static int cmd_merge(int argc, char **argv)
{
	int threads = hw_threads(), level = -1; std::vector<const char*> files;
	for (int i = 0; i < argc; ++i) {
		if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-l") && i + 1 < argc) level = atoi(argv[++i]);
		else if (argv[i][0] == '-' && argv[i][1]) die(std::string("merge: unsupported option ") + argv[i]);
		else files.push_back(argv[i]);
	}
	if (files.size() < 2) die("usage: sambamba merge [-t N] out.bam in1.bam [in2.bam ...]");
	std::vector<merge_src_t> src(files.size() - 1);
	for (size_t i = 0; i < src.size(); ++i) { src[i].fd = open_in(files[i + 1]); src[i].in.reset(new bgzf_in_t(src[i].fd, 2)); if (!hdr_read(*src[i].in, src[i].h)) die(std::string("merge: not a BAM file: ") + files[i + 1]); }
	/* header: the first file's, plus the @RG / @PG / @CO lines of the others that it does not hold yet; the references must agree */
	bam_hdr_t h = src[0].h;
	for (size_t i = 1; i < src.size(); ++i) {
		if (src[i].h.names != h.names || src[i].h.lens != h.lens) die("merge: the inputs have different reference sequences");
		size_t p = 0; const std::string &t = src[i].h.text;
		while (p < t.size()) {
			size_t e = t.find('\n', p); if (e == std::string::npos) e = t.size();
			const std::string line = t.substr(p, e - p);
			if (line.size() > 3 && (!line.compare(0, 3, "@RG") || !line.compare(0, 3, "@PG") || !line.compare(0, 3, "@CO")) && ("\n" + h.text).find("\n" + line + "\n") == std::string::npos) h.text += line + "\n";
			p = e + 1;
		}
	}
	int ofd = open(files[0], O_WRONLY | O_CREAT | O_TRUNC, 0644); if (ofd < 0) die(std::string("merge: cannot write ") + files[0]);
	bgzf_out_t out(ofd, level, threads); hdr_write(out, h);
	kway_merge(src, out); out.finish(); close(ofd);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: sambamba <view|sort|index|merge> ...  (libssgpu %s, %s)\n", ssg_version(), ssg_backend()); return 1; }
#ifdef F_SETPIPE_SZ
	(void)fcntl(0, F_SETPIPE_SZ, 1 << 20); (void)fcntl(1, F_SETPIPE_SZ, 1 << 20);   /* the reference's pipelines: fewer wake-ups per megabyte (fails harmlessly on files) */
#endif
	if (!strcmp(argv[1], "view")) return cmd_view(argc - 2, argv + 2);
	if (!strcmp(argv[1], "sort")) { ssg_worker_begin(); ssg_stamp("sambamba_sort", "start"); const int rc = cmd_sort(argc - 2, argv + 2); ssg_stamp("sambamba_sort", "end"); ssg_worker_done(rc); return ssg_fast_exit(rc); }
	if (!strcmp(argv[1], "index")) { ssg_stamp("sambamba_index", "start"); const int rc = cmd_index(argc - 2, argv + 2); ssg_stamp("sambamba_index", "end"); return rc; }
	if (!strcmp(argv[1], "flagstat")) return cmd_flagstat(argc - 2, argv + 2);
	if (!strcmp(argv[1], "merge")) return cmd_merge(argc - 2, argv + 2);
	if (!strcmp(argv[1], "index-parts")) return cmd_index_parts(argc - 2, argv + 2);
	fprintf(stderr, "[sambamba] unsupported sub-command %s\n", argv[1]);
	return 2;
}
