/*
 * ranks.h -- rank mode: several pipelines of the reference's script side by side, one per GPU (bin/speedseq-ranks).
 *
 * The reference has one pipeline `bwa mem | samblaster | sambamba view | sambamba sort` (/root/reference/bin/speedseq:438-441); its results
 * depend on the input order in three places: upstream's batches (the scope of the insert-size model), samblaster's first-seen-wins duplicate
 * set, and the order of equal sort keys.  Rank r of SSG_WORLD pipelines aligns the batches r, r + N, ... (bwa_main.cpp; ranksplit.h: who reads
 * which bytes); the samblasters share ONE duplicate set, sharded over them by signature -- every rank serves a slice of the table and asks all
 * of them, each slice decided in batch order (samblaster_main.cpp); side-stream lines travel to rank 0 and leave it in batch order, so that its
 * two small sorts see what a single pipeline's would; the main stream's records carry (batch, index) ordinals into the sorts, which exchange
 * sorted runs through files, merge one stretch of the genome each and place it in the one output file (sambamba_main.cpp).  Everything between
 * the processes goes through SSG_RDV (and SSG_RDV_DATA for the bulk): UNIX sockets for the samblasters, files for the sorts.
 */
#ifndef SSG_RANKS_H
#define SSG_RANKS_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <unistd.h>
#include <time.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <dirent.h>
#include <fcntl.h>
#include <string>
#include <vector>

static inline int rk_world() { const char *e = getenv("SSG_WORLD"); return e && atoi(e) > 1 ? atoi(e) : 1; }
static inline int rk_rank() { const char *e = getenv("SSG_RANK"); return e ? atoi(e) : 0; }
static inline std::string rk_dir() { const char *e = getenv("SSG_RDV"); return e ? e : ""; }
/* where the ranks' BULK data goes -- the sorts' exchange runs with their ordinals, the batches of a served (compressed) input: SSG_RDV_DATA, a
 * directory on disk that bin/speedseq-ranks makes next to the output (a whole genome's records do not belong in a memory file system);
 * SSG_RDV itself keeps the socket and the small marker files */
static inline std::string rk_data_dir() { const char *e = getenv("SSG_RDV_DATA"); return e && *e ? e : rk_dir(); }
static inline bool rk_check(const char *who)
{
	if (rk_world() == 1) return true;
	if (rk_rank() < 0 || rk_rank() >= rk_world() || rk_dir().empty()) { fprintf(stderr, "[%s] rank mode needs SSG_RANK in 0 .. SSG_WORLD - 1 and SSG_RDV (a directory all ranks share)\n", who); return false; }
	return true;
}
static inline double rk_now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static inline double rk_timeout() { const char *e = getenv("SSG_RDV_TIMEOUT"); return e && atof(e) > 0 ? atof(e) : 86400.0; }   /* seconds a rank waits for another (tests set it low) */

/* a stage that gives up in rank mode says so in the rendezvous directory: the reference's script does not stop when a stage of its pipeline
 * fails, and the other ranks would wait for this one until their patience (SSG_RDV_TIMEOUT) ends */
static inline void rk_mark_failed(const char *who)
{
	if (rk_world() == 1 || rk_dir().empty()) return;
	const std::string p = rk_dir() + "/failed." + std::to_string(rk_rank()) + "." + who;
	const int fd = open(p.c_str(), O_WRONLY | O_CREAT, 0644); if (fd >= 0) close(fd);
}
static inline bool rk_someone_failed()
{
	DIR *d = opendir(rk_dir().c_str()); if (!d) return false;
	bool f = false;
	while (struct dirent *e = readdir(d)) if (!strncmp(e->d_name, "failed.", 7)) { f = true; break; }
	closedir(d); return f;
}

enum { RK_ENDS = 1, RK_DUP = 2, RK_SIDE = 3, RK_DONE = 4 };
struct rk_hdr_t { uint32_t type, rank; uint64_t b, len; };

static inline bool rk_write_all(int fd, const void *p, size_t n)
{
	const uint8_t *b = (const uint8_t*)p;
	while (n) { const ssize_t w = send(fd, b, n, MSG_NOSIGNAL); if (w < 0) { if (errno == EINTR) continue; return false; } b += w; n -= (size_t)w; }
	return true;
}
static inline bool rk_read_all(int fd, void *p, size_t n)
{
	uint8_t *b = (uint8_t*)p;
	while (n) { const ssize_t r = recv(fd, b, n, 0); if (r < 0) { if (errno == EINTR) continue; return false; } if (r == 0) return false; b += r; n -= (size_t)r; }
	return true;
}
/* one message: header, then the parts of the payload one after the other */
static inline bool rk_send(int fd, uint32_t type, int rank, uint64_t b, const void *p0, size_t n0, const void *p1 = 0, size_t n1 = 0, const void *p2 = 0, size_t n2 = 0)
{
	rk_hdr_t h; h.type = type; h.rank = (uint32_t)rank; h.b = b; h.len = (uint64_t)(n0 + n1 + n2);
	return rk_write_all(fd, &h, sizeof(h)) && (!n0 || rk_write_all(fd, p0, n0)) && (!n1 || rk_write_all(fd, p1, n1)) && (!n2 || rk_write_all(fd, p2, n2));
}
static inline bool rk_recv(int fd, rk_hdr_t &h, std::vector<uint8_t> &payload)
{
	if (!rk_read_all(fd, &h, sizeof(h)) || h.len > ((uint64_t)1 << 40)) return false;
	payload.resize((size_t)h.len);
	return !h.len || rk_read_all(fd, payload.data(), (size_t)h.len);
}
static inline int rk_listen(const std::string &path, int backlog)
{
	sockaddr_un a; memset(&a, 0, sizeof(a)); a.sun_family = AF_UNIX;
	if (path.size() >= sizeof(a.sun_path)) { fprintf(stderr, "[ranks] SSG_RDV is too long a path for a socket\n"); return -1; }
	strcpy(a.sun_path, path.c_str()); unlink(path.c_str());
	const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
	if (fd < 0 || bind(fd, (sockaddr*)&a, sizeof(a)) != 0 || listen(fd, backlog) != 0) { perror("[ranks] listen"); if (fd >= 0) close(fd); return -1; }
	return fd;
}
static inline int rk_connect(const std::string &path)
{
	sockaddr_un a; memset(&a, 0, sizeof(a)); a.sun_family = AF_UNIX;
	if (path.size() >= sizeof(a.sun_path)) return -1;
	strcpy(a.sun_path, path.c_str());
	const double t0 = rk_now();
	for (;;) {
		const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
		if (fd < 0) return -1;
		if (connect(fd, (sockaddr*)&a, sizeof(a)) == 0) return fd;
		close(fd);
		if (rk_now() - t0 > rk_timeout()) { fprintf(stderr, "[ranks] nobody listens at %s\n", path.c_str()); return -1; }
		if (rk_someone_failed()) { fprintf(stderr, "[ranks] another rank's pipeline failed\n"); return -1; }
		usleep(20000);
	}
}
/* files as messages between the sorts: written under a temporary name, renamed into place; a reader polls for the name */
static inline bool rk_file_wait(const std::string &path)
{
	const double t0 = rk_now(); struct stat sb;
	while (stat(path.c_str(), &sb) != 0) {
		if (rk_now() - t0 > rk_timeout()) { fprintf(stderr, "[ranks] %s did not appear\n", path.c_str()); return false; }
		if (rk_someone_failed()) { fprintf(stderr, "[ranks] another rank's pipeline failed\n"); return false; }
		usleep(20000);
	}
	return true;
}
static inline bool rk_file_put(const std::string &path, const void *p, size_t n)
{
	const std::string tmp = path + ".tmp";
	FILE *f = fopen(tmp.c_str(), "wb"); if (!f) return false;
	const bool ok = (!n || fwrite(p, 1, n, f) == n) && fclose(f) == 0;
	return ok && rename(tmp.c_str(), path.c_str()) == 0;
}
static inline bool rk_file_get(const std::string &path, std::vector<uint8_t> &out)
{
	FILE *f = fopen(path.c_str(), "rb"); if (!f) return false;
	fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
	out.resize((size_t)(n > 0 ? n : 0));
	const bool ok = !n || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
	fclose(f); return ok;
}
#endif
