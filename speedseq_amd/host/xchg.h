/*
 * xchg.h -- rank mode's exchange step between the `sambamba sort` of every rank (SURVEY.md 8e coupling 3): an all-to-all of variable-size
 * blocks -- rank r sends rank q the records of q's stretch of the genome -- behind one interface with two transports:
 *   rccl   libssgpu's ssg_coll_* (csrc/ssg_coll.cpp): grouped ncclSend / ncclRecv over xGMI, one process per GPU; the product's transport on a node of MI355X;
 *   sock   UNIX sockets in SSG_RDV, a thread per peer and direction: what the host emulation runs (no RCCL there) and what a node without peer access falls to.
 * Rounds 4-5 exchanged sorted runs through files; that path is still the one for inputs beyond memory and when no transport comes up (SSG_RANKS_XCHG=file).
 */
#ifndef SSG_XCHG_H
#define SSG_XCHG_H
#include <thread>
#include <atomic>
#include <memory>
#include <vector>
#include <string>
#include "ranks.h"
#include "../../include/ssgpu.h"

struct xchg_t {
	int rank, world;
	xchg_t(int r, int w) : rank(r), world(w) {}
	virtual ~xchg_t() {}
	virtual const char *name() const = 0;
	/* block q of send (send_bytes[q]) to rank q; recv[q] receives recv_bytes[q] from rank q (sizes agreed beforehand) */
	virtual bool alltoallv(const void *const *send, const uint64_t *send_bytes, void *const *recv, const uint64_t *recv_bytes) = 0;
	bool alltoall_u64(const uint64_t *send, uint64_t *recv, int k)
	{
		std::vector<const void*> sp((size_t)world); std::vector<void*> rp((size_t)world); std::vector<uint64_t> n((size_t)world, (uint64_t)k * 8);
		for (int q = 0; q < world; ++q) { sp[(size_t)q] = send + (size_t)q * k; rp[(size_t)q] = recv + (size_t)q * k; }
		return alltoallv(sp.data(), n.data(), rp.data(), n.data());
	}
};

struct xchg_rccl_t : xchg_t {
	ssg_coll_t *c;
	xchg_rccl_t(int r, int w) : xchg_t(r, w), c(0) {}
	bool up(const std::string &rdv) { return ssg_coll_init(rank, world, rdv.c_str(), &c) == 0; }
	~xchg_rccl_t() { ssg_coll_destroy(c); }
	const char *name() const { return "RCCL (grouped ncclSend / ncclRecv)"; }
	bool alltoallv(const void *const *send, const uint64_t *send_bytes, void *const *recv, const uint64_t *recv_bytes)
	{
		if (ssg_coll_alltoallv(c, send, send_bytes, recv, recv_bytes)) { fprintf(stderr, "[sambamba] sort: exchange over RCCL failed: %s\n", ssg_last_error()); return false; }
		return true;
	}
};

struct xchg_sock_t : xchg_t {
	std::vector<int> out_fd, in_fd; int lfd;
	xchg_sock_t(int r, int w) : xchg_t(r, w), out_fd((size_t)w, -1), in_fd((size_t)w, -1), lfd(-1) {}
	~xchg_sock_t() { for (int fd : out_fd) if (fd >= 0) close(fd); for (int fd : in_fd) if (fd >= 0) close(fd); if (lfd >= 0) close(lfd); }
	const char *name() const { return "UNIX sockets"; }
	bool up(const std::string &rdv)
	{
		lfd = rk_listen(rdv + "/sort." + std::to_string(rank) + ".sock", world + 2);
		if (lfd < 0) return false;
		bool ok = true;
		std::thread acc([&]() {   /* every other rank connects once and says who it is */
			for (int k = 0; k + 1 < world && ok; ++k) {
				const int fd = accept(lfd, 0, 0);
				uint32_t who = 0;
				if (fd < 0 || !rk_read_all(fd, &who, 4) || who >= (uint32_t)world || in_fd[who] >= 0) { ok = false; if (fd >= 0) close(fd); break; }
				in_fd[who] = fd;
			}
		});
		for (int q = 0; q < world; ++q) if (q != rank) {
			out_fd[(size_t)q] = rk_connect(rdv + "/sort." + std::to_string(q) + ".sock");
			const uint32_t me = (uint32_t)rank;
			if (out_fd[(size_t)q] < 0 || !rk_write_all(out_fd[(size_t)q], &me, 4)) { ok = false; break; }
		}
		if (!ok) { shutdown(lfd, SHUT_RDWR); }   /* (lets a blocked accept go) */
		acc.join();
		return ok;
	}
	bool alltoallv(const void *const *send, const uint64_t *send_bytes, void *const *recv, const uint64_t *recv_bytes)
	{
		std::vector<std::thread> th; std::vector<int> bad((size_t)(2 * world), 0);
		for (int q = 0; q < world; ++q) if (q != rank) {
			th.emplace_back([&, q]() { if (send_bytes[q] && !rk_write_all(out_fd[(size_t)q], send[q], (size_t)send_bytes[q])) bad[(size_t)q] = 1; });
			th.emplace_back([&, q]() { if (recv_bytes[q] && !rk_read_all(in_fd[(size_t)q], recv[q], (size_t)recv_bytes[q])) bad[(size_t)(world + q)] = 1; });
		}
		if (send_bytes[rank] != recv_bytes[rank]) bad[0] = 1; else if (send_bytes[rank]) memcpy(recv[rank], send[rank], (size_t)send_bytes[rank]);
		for (std::thread &x : th) x.join();
		for (int b : bad) if (b) { fprintf(stderr, "[sambamba] sort: exchange over sockets failed (a rank is gone)\n"); return false; }
		return true;
	}
};

/* SSG_RANKS_XCHG: rccl | sock | file (bin/speedseq-ranks sets rccl when every rank has a device of its own; unset = file).  NULL = the files of rounds 4-5.
 * Which transport is used is agreed through the rendezvous directory before any collective call is made (and again after the first one): a rank that cannot
 * bring RCCL up must not leave the others waiting inside a collective call. */
static inline xchg_t *xchg_open(int rank, int world, const std::string &rdv)
{
	const char *e = getenv("SSG_RANKS_XCHG");
	const std::string want = e ? e : "file";
	if (want == "file") return 0;
	if (want == "sock") {
		xchg_sock_t *s = new xchg_sock_t(rank, world);
		if (s->up(rdv)) return s;
		delete s;
		fprintf(stderr, "[sambamba] sort: the socket transport did not come up\n");   /* the ranks cannot agree on anything else by now: this run fails */
		return (xchg_t*)-1;
	}
	const char can = ssg_coll_available() ? 'y' : 'n';
	if (!rk_file_put(rdv + "/xchg." + std::to_string(rank), &can, 1)) return 0;
	bool all = can == 'y';
	for (int q = 0; q < world; ++q) { std::vector<uint8_t> b; if (!rk_file_wait(rdv + "/xchg." + std::to_string(q)) || !rk_file_get(rdv + "/xchg." + std::to_string(q), b) || b.size() != 1 || b[0] != 'y') all = false; }
	if (!all) { if (rank == 0) fprintf(stderr, "[sambamba] sort: RCCL is not available on every rank; exchanging through files\n"); return 0; }
	/* The communicator and a first exchange of one word with every peer (RCCL sets its peer connections up on first use) are made by a thread that is given
	 * SSG_RANKS_RCCL_TIMEOUT seconds (90); what came of it is agreed through the directory once more, so that one rank's failure sends ALL ranks to the files
	 * instead of leaving the others inside a collective call.  A communicator that is not used is left alone (destroying it can block on the rank that failed). */
	struct att_t { std::atomic<int> st{0}; xchg_rccl_t *r = 0; std::string why; };
	std::shared_ptr<att_t> att = std::make_shared<att_t>();
	att->r = new xchg_rccl_t(rank, world);
	std::thread([att, rdv, world]() {
		bool ok = att->r->up(rdv);
		if (ok) { std::vector<uint64_t> a((size_t)world, 1), b((size_t)world, 0); ok = att->r->alltoall_u64(a.data(), b.data(), 1); for (uint64_t v : b) if (v != 1) ok = false; }
		if (!ok) att->why = ssg_last_error();
		att->st.store(ok ? 1 : 2, std::memory_order_release);
	}).detach();
	const char *te = getenv("SSG_RANKS_RCCL_TIMEOUT");
	const double limit = te && atof(te) > 0 ? atof(te) : 90.0;
	double waited = 0;
	while (att->st.load(std::memory_order_acquire) == 0 && waited < limit) { usleep(5000); waited += 0.005; }
	const int st = att->st.load(std::memory_order_acquire);
	const char up = st == 1 ? 'y' : 'n';
	if (st != 1) fprintf(stderr, "[sambamba] sort: rank %d has no RCCL exchange (%s)\n", rank, st == 0 ? "no answer within the time limit" : att->why.c_str());
	if (!rk_file_put(rdv + "/xchgup." + std::to_string(rank), &up, 1)) return (xchg_t*)-1;
	all = up == 'y';
	for (int q = 0; q < world; ++q) { std::vector<uint8_t> b; if (!rk_file_wait(rdv + "/xchgup." + std::to_string(q)) || !rk_file_get(rdv + "/xchgup." + std::to_string(q), b) || b.size() != 1 || b[0] != 'y') all = false; }
	if (all) return att->r;
	if (rank == 0) fprintf(stderr, "[sambamba] sort: RCCL did not come up on every rank; exchanging through files\n");
	return 0;
}
#endif
