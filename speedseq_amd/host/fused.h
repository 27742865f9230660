/*
 * fused.h -- the binary hand-off between this repository's `bwa mem`, `samblaster` and `sambamba` (SURVEY.md 7.1 "fused mode").
 *
 * The reference wires its stages with SAM text on pipes (/root/reference/bin/speedseq:438-441); at MI355X alignment rates the text
 * (0.8 GB/s to print, parse, re-print and parse again) is the wall.  When speedseq.config -- which the script `source`s, so the
 * variable reaches every stage -- exports SSG_FUSED=1, `bwa mem` writes the records it would have printed as BAM records (the bytes
 * `sambamba view -S -f bam` makes of its lines) in frames, `samblaster` takes its decisions on those records (same device entry point,
 * same options, the line view derived from the BAM fields instead of the text), patches FLAG / MC / MQ and forwards them,
 * `sambamba view` passes the frames through and `sambamba sort` reads them directly.  The side streams stay SAM text (the script
 * pipes them through gawk): bwa attaches the text of the few pairs that can qualify (supplementary lines, or both ends mapped
 * and not a proper pair), samblaster emits from that text exactly as in text mode.  The text path remains the parity path; both must
 * produce the same three BAM files (tests/test_fused.py).  `bwa mem -C` (speedseq realign: another program sits between bwa and
 * samblaster there) never fuses.
 *
 * Stream: 8 magic bytes, then frames { u32 type, u32 0, u64 payload bytes }.
 *   HEADER  SAM header text (every stage may append its @PG line)
 *   BATCH   bwa -> samblaster: fu_batch_t, candidate table, candidate text, BAM records
 *   MAIN    samblaster -> sort: BAM records (block_size-prefixed, htslib sam.c:443-473)
 *   END     no payload
 *   REF     the payload of a BATCH / MAIN frame lives in a file on a memory file system instead of travelling through the pipe:
 *           fu_ref_t (the frame's own type and payload size) followed by the path.  The writer fills the mapped file with several
 *           threads and sends only this note; the reader maps the file and unlinks it at once, so the pages go away with the last
 *           mapping.  A pipe moves ~1 GB/s through one thread on each side, and the reference's pipeline has three of them in a
 *           row on the main stream; at several hundred MB per device call that was the pace of the whole fused pipeline.
 *           Directory: SSG_FUSED_SHM (default /dev/shm; "0" or an unusable directory = payloads through the pipe as before).
 */
#ifndef SSG_FUSED_H
#define SSG_FUSED_H
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <errno.h>
#include <unistd.h>
#include <stdio.h>
#include <fcntl.h>
#include <time.h>
#include <signal.h>
#include <dirent.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <algorithm>
/* cores this process may really use: the cgroup's CPU quota when there is one (a container next to the GPU: 256 hardware threads visible, 16 cores granted), else the
 * hardware threads.  A host pool larger than this only takes the quota away from the threads that feed the device: at 200 M pairs the sort's 128 gather threads held
 * `bwa mem`'s device calls to half their rate (profiles/r06_soak_200M.json) */
static inline double ssg_usable_cores()
{
	double c = (double)std::max(1u, std::thread::hardware_concurrency());
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[64]; double per = 0; if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) c = std::min(c, atof(q) / per); fclose(f); }
	return c;
}
/* a pool of `asked` threads on this host: no more than the usable cores (rounded up, at least 4); SSG_POOL_CAP=0: as asked */
static inline int ssg_pool_threads(int asked)
{
	const char *e = getenv("SSG_POOL_CAP");
	if (e && atoi(e) == 0) return std::max(1, asked);
	return std::max(1, std::min(asked, std::max(4, (int)(ssg_usable_cores() + 0.999))));
}
#include <sys/statvfs.h>
#include <string>
#include <atomic>

/* SSG_STAMP=1: every stage of the pipeline says when it started and ended, in seconds of the realtime clock (bench.py / tools/dbg/literal_ab.py lay them on one axis) */
#include <time.h>
static inline void ssg_stamp(const char *who, const char *what)
{
	static const bool on = getenv("SSG_STAMP") != 0;
	if (!on) return;
	struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
	fprintf(stderr, "[stamp] %s %s %.3f\n", who, what, (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
}
/* the end of a stage's main(): everything it owed is written and closed; what is left is giving gigabytes of page-locked blocks, HBM arenas and mapped
 * frames back one by one (destructors, the HIP runtime's exit handlers) -- the kernel does that wholesale when the process is gone.  SSG_FAST_EXIT=0: the long way. */
#include <unistd.h>
static inline int ssg_fast_exit(int rc)
{
	const char *e = getenv("SSG_FAST_EXIT");
	if (e && !strcmp(e, "0")) return rc;
	fflush(stdout); fflush(stderr);
	_exit(rc);
}
/* A stage whose LAST act is long -- `sambamba sort` ends holding gigabytes of mapped frames, page-locked blocks and a GPU context, and the kernel needs
 * 0.4 - 0.9 s to take all that back when the process goes (measured in round 6 with SSG_STAMP: `written' to `exited') -- runs as a worker behind a
 * waiter: the process the pipeline started forks before it has a thread or a device, the child does the stage's work and says so through a pipe when the
 * output is complete and closed, the waiter exits with that status at once and the script goes on while the worker is taken apart.  A worker that dies
 * without saying anything is waited for and its status passed on.  SSG_DETACH=0: one process, as ever. */
#include <sys/wait.h>
static int ssg_worker_fd = -1;
static inline void ssg_worker_begin()
{
	const char *e = getenv("SSG_DETACH");
	if (e && !strcmp(e, "0")) return;
	int pfd[2];
	if (pipe(pfd) != 0) return;
	const pid_t c = fork();
	if (c < 0) { close(pfd[0]); close(pfd[1]); return; }
	if (c == 0) { close(pfd[0]); ssg_worker_fd = pfd[1]; return; }   /* the worker: goes on as the stage */
	close(pfd[1]);
	unsigned char st = 0; ssize_t n;
	do n = read(pfd[0], &st, 1); while (n < 0 && errno == EINTR);
	if (n == 1) _exit(st);
	int ws = 0; pid_t w;
	do w = waitpid(c, &ws, 0); while (w < 0 && errno == EINTR);
	_exit(w == c && WIFEXITED(ws) ? WEXITSTATUS(ws) : 1);
}
/* the worker's output is complete: the waiter may go.  The worker's own descriptors are closed first -- whoever reads the pipeline's stderr must not wait for the teardown */
static inline void ssg_worker_done(int rc)
{
	if (ssg_worker_fd < 0) return;
	fflush(stdout); fflush(stderr);
	const int fd = ssg_worker_fd; ssg_worker_fd = -1;
	for (int k = 0; k < 3; ++k) close(k);
	const unsigned char st = (unsigned char)rc;
	ssize_t n; do n = write(fd, &st, 1); while (n < 0 && errno == EINTR);
	close(fd);
}
#define FU_MAGIC "SSGFUSE1"
enum { FU_HEADER = 1, FU_BATCH = 2, FU_MAIN = 3, FU_END = 4, FU_REF = 5 };
struct fu_frame_t { uint32_t type, zero; uint64_t len; };
struct fu_ref_t { uint32_t type, zero; uint64_t len; };   /* REF payload: this, then the path of the file holding the `len` payload bytes */
#define FU_SEG_PREFIX "ssgfuse."
static inline size_t fu_seg_min() { const char *e = getenv("SSG_FUSED_SHM_MIN"); return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)1 << 20; }   /* smaller payloads go through the pipe */
/* BATCH payload: fu_batch_t | fu_cand_t[n_cand] | text[text_bytes] | bam[bam_bytes].  A candidate names its first record (ordinal within
 * the batch) and owns text[text_off .. next candidate's text_off): the SAM lines of those records, one per record, in order. */
struct fu_batch_t { uint64_t n_rec, bam_bytes, n_cand, text_bytes; };
struct fu_cand_t { uint64_t first_rec, n_rec, text_off; };

static inline bool fu_enabled() { const char *e = getenv("SSG_FUSED"); return e && atoi(e) != 0; }
static inline bool fu_read_full(int fd, void *buf, size_t n)
{
	uint8_t *b = (uint8_t*)buf;
	while (n) { ssize_t r = read(fd, b, n); if (r < 0) { if (errno == EINTR) continue; return false; } if (r == 0) return false; b += r; n -= (size_t)r; }
	return true;
}
static inline bool fu_write_full(int fd, const void *buf, size_t n)
{
	const uint8_t *b = (const uint8_t*)buf;
	while (n) { ssize_t w = write(fd, b, n); if (w < 0) { if (errno == EINTR) continue; return false; } b += w; n -= (size_t)w; }
	return true;
}
static inline bool fu_write_frame(int fd, uint32_t type, const void *payload, uint64_t len)
{
	fu_frame_t f; f.type = type; f.zero = 0; f.len = len;
	return fu_write_full(fd, &f, sizeof(f)) && (!len || fu_write_full(fd, payload, (size_t)len));
}

/* payload of one frame: heap memory (capacity kept when reused) or a mapped segment */
struct fu_buf_t {
	uint8_t *p; size_t len, cap; bool mapped;
	fu_buf_t() : p(0), len(0), cap(0), mapped(false) {}
	fu_buf_t(fu_buf_t &&o) : p(o.p), len(o.len), cap(o.cap), mapped(o.mapped) { o.p = 0; o.len = o.cap = 0; o.mapped = false; }
	fu_buf_t &operator=(fu_buf_t &&o) { if (this != &o) { reset(); p = o.p; len = o.len; cap = o.cap; mapped = o.mapped; o.p = 0; o.len = o.cap = 0; o.mapped = false; } return *this; }
	fu_buf_t(const fu_buf_t&) = delete; fu_buf_t &operator=(const fu_buf_t&) = delete;
	~fu_buf_t() { reset(); }
	void reset() { if (p) { if (mapped) munmap(p, cap); else free(p); } p = 0; len = cap = 0; mapped = false; }
	bool heap(size_t n)                                   /* n bytes of heap memory, contents undefined */
	{
		if (mapped) reset();
		if (cap < n) { free(p); cap = n + n / 8; p = (uint8_t*)malloc(cap ? cap : 1); if (!p) { cap = 0; return false; } }
		len = n; return true;
	}
};

/* directory for segments, or NULL when they are switched off */
static inline const char *fu_seg_dir()
{
	const char *e = getenv("SSG_FUSED_SHM");
	if (!e) return "/dev/shm";
	return (*e && strcmp(e, "0")) ? e : 0;
}
/* a fresh segment of `len` bytes, mapped for writing; false = send the payload through the pipe instead */
static inline bool fu_seg_create(size_t len, fu_buf_t &b, std::string &path)
{
	const char *dir = fu_seg_dir();
	if (!dir || len < fu_seg_min()) return false;
	struct statvfs vs;
	if (statvfs(dir, &vs) != 0 || (double)vs.f_bavail * (double)vs.f_frsize < 2.0 * (double)len + 1073741824.0) return false;   /* a full tmpfs is a SIGBUS, not an error code */
	static std::atomic<unsigned long> seq(0);
	char nm[96]; snprintf(nm, sizeof(nm), "/" FU_SEG_PREFIX "%ld.%lu", (long)getpid(), seq.fetch_add(1));
	path = std::string(dir) + nm;
	const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_CLOEXEC, 0600);
	if (fd < 0) return false;
	/* the pages are reserved here, in one call (measured: 0.07 s + 0.07 s to fill 670 MB with 8 threads, against 0.14-0.55 s when every
	 * first touch allocates), and a file system without room says so now instead of with a SIGBUS in the middle of the copy */
	bool room = ftruncate(fd, (off_t)len) == 0;
	if (room && fallocate(fd, 0, 0, (off_t)len) != 0 && errno != EOPNOTSUPP && errno != ENOSYS) room = false;
	void *m = room ? mmap(0, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
	close(fd);
	if (m == MAP_FAILED) { unlink(path.c_str()); return false; }
	b.reset(); b.p = (uint8_t*)m; b.len = b.cap = len; b.mapped = true;
	return true;
}
/* hand a filled segment to the reader of `fd` as a frame of `type`; the mapping is released here */
static inline bool fu_seg_send(int fd, uint32_t type, fu_buf_t &b, const std::string &path)
{
	fu_ref_t r; r.type = type; r.zero = 0; r.len = b.len;
	std::string pl((const char*)&r, sizeof(r)); pl += path;
	b.reset();
	if (fu_write_frame(fd, FU_REF, pl.data(), pl.size())) return true;
	unlink(path.c_str());
	return false;
}
/* next frame of the stream: h = its type and payload size, b = the payload (heap, or the mapped segment of a REF frame) */
static inline bool fu_read_frame(int fd, fu_frame_t &h, fu_buf_t &b)
{
	if (!fu_read_full(fd, &h, sizeof(h))) return false;
	if (h.type != FU_REF) { if (!b.heap((size_t)h.len)) return false; return !h.len || fu_read_full(fd, b.p, (size_t)h.len); }
	if (h.len <= sizeof(fu_ref_t) || h.len > sizeof(fu_ref_t) + 4096) return false;
	std::string pl((size_t)h.len, '\0');
	if (!fu_read_full(fd, &pl[0], pl.size())) return false;
	fu_ref_t r; memcpy(&r, pl.data(), sizeof(r));
	const std::string path = pl.substr(sizeof(r));
	const size_t sl = path.rfind('/');
	if (path.compare(sl == std::string::npos ? 0 : sl + 1, strlen(FU_SEG_PREFIX), FU_SEG_PREFIX) != 0 || r.type == FU_REF) return false;   /* only files this protocol made are opened and unlinked */
	const int sfd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
	if (sfd < 0) return false;
	struct stat sb;
	void *m = (fstat(sfd, &sb) == 0 && (uint64_t)sb.st_size >= r.len && r.len) ? mmap(0, (size_t)r.len, PROT_READ, MAP_SHARED, sfd, 0) : MAP_FAILED;
	close(sfd); unlink(path.c_str());
	if (m == MAP_FAILED) return false;
	b.reset(); b.p = (uint8_t*)m; b.len = b.cap = (size_t)r.len; b.mapped = true;
	h.type = r.type; h.len = r.len;
	return true;
}
/* after a broken frame: unlink the segments the rest of the stream names (best effort; stops at the first thing that is not a frame) */
static inline void fu_discard_rest(int fd)
{
	std::string pl; char sink[65536];
	for (;;) {
		fu_frame_t h;
		if (!fu_read_full(fd, &h, sizeof(h)) || h.type < FU_HEADER || h.type > FU_REF || h.zero) return;
		if (h.type == FU_REF) {
			if (h.len <= sizeof(fu_ref_t) || h.len > sizeof(fu_ref_t) + 4096) return;
			pl.resize((size_t)h.len);
			if (!fu_read_full(fd, &pl[0], pl.size())) return;
			const std::string path = pl.substr(sizeof(fu_ref_t));
			const size_t sl = path.rfind('/');
			if (path.compare(sl == std::string::npos ? 0 : sl + 1, strlen(FU_SEG_PREFIX), FU_SEG_PREFIX) == 0) unlink(path.c_str());
		} else for (uint64_t left = h.len; left; ) { const size_t k = (size_t)(left < sizeof(sink) ? left : sizeof(sink)); if (!fu_read_full(fd, sink, k)) return; left -= k; }
		if (h.type == FU_END) return;
	}
}
/* segments left behind by a pipeline that died between a writer's send and its reader's open: removed when their writer is gone and
 * they are older than an hour (a live pipeline's writer may exit before its last segments are read) */
static inline void fu_seg_sweep()
{
	const char *dir = fu_seg_dir(); if (!dir) return;
	DIR *d = opendir(dir); if (!d) return;
	const time_t now = time(0);
	while (struct dirent *e = readdir(d)) {
		if (strncmp(e->d_name, FU_SEG_PREFIX, strlen(FU_SEG_PREFIX)) != 0) continue;
		const long pid = atol(e->d_name + strlen(FU_SEG_PREFIX));
		const std::string p = std::string(dir) + "/" + e->d_name;
		struct stat sb;
		if (pid > 0 && stat(p.c_str(), &sb) == 0 && now - sb.st_mtime > 3600 && kill((pid_t)pid, 0) != 0 && errno == ESRCH) unlink(p.c_str());
	}
	closedir(d);
}
#endif
