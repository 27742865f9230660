/*
 * ssg_coll.cpp -- the one exchange step of the path over several GPUs: the coordinate-sorted merge (SURVEY.md 8e coupling 3; the reference ends in ONE
 * OUT.bam, /root/reference/bin/speedseq:441,491-495).  Rank mode runs a pipeline per GPU (bin/speedseq-ranks); every rank's `sambamba sort` holds the records
 * of its rank's batches and owns a stretch of the genome, so each sends every other the records of that rank's stretch: an all-to-all of variable-size blocks.
 * Here that is grouped ncclSend / ncclRecv over RCCL -- all seven xGMI links of a device busy at once, no ring -- on blocks staged through HBM (the records
 * are made by host stages, so a block goes host -> HBM -> peer's HBM -> peer's host; the rendezvous files of rounds 4-5 stay as the fall-back and for
 * inputs beyond memory).  RCCL is loaded on first use (dlopen): nothing else in the library needs it.  One process per GPU: the communicator is made
 * over the process's current device; the unique id travels through the rendezvous directory.
 */
#include <string>
#include <vector>
#include <unistd.h>
#include <sys/stat.h>
#include "ssg_rt.h"
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"
#ifndef SSG_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

SSG_ABI_FP_DEFINE(coll)
#define CHK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

#ifndef SSG_EMU
struct ssg_rccl_api {
	void *lib = 0;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = 0;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = 0;
	ncclResult_t (*CommDestroy)(ncclComm_t) = 0;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = 0;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = 0;
	ncclResult_t (*GroupStart)() = 0;
	ncclResult_t (*GroupEnd)() = 0;
	const char *(*GetErrorString)(ncclResult_t) = 0;
	bool load()
	{
		if (lib) return true;
		for (const char *n : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
		if (!lib) return false;
#define SSG_SYM(f) do { *(void**)&f = dlsym(lib, "nccl" #f); if (!f) return false; } while (0)
		SSG_SYM(GetUniqueId); SSG_SYM(CommInitRank); SSG_SYM(CommDestroy); SSG_SYM(Send); SSG_SYM(Recv); SSG_SYM(GroupStart); SSG_SYM(GroupEnd); SSG_SYM(GetErrorString);
#undef SSG_SYM
		return true;
	}
};
static ssg_rccl_api rccl;
#define NCHK(x, what) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { ssg_err_msg = std::string(what) + ": " + rccl.GetErrorString(r_); return SSG_EHIP; } } while (0)
#endif

struct ssg_coll {
	int rank = 0, world = 1;
#ifndef SSG_EMU
	ncclComm_t comm = 0; hipStream_t st = 0;
#endif
};

extern "C" {

/* 1 when this process has a device and RCCL can be loaded: what the ranks tell one another before any of them enters ncclCommInitRank */
int ssg_coll_available(void)
{
#ifdef SSG_EMU
	return 0;
#else
	return rt_device_count() >= 1 && rccl.load() ? 1 : 0;
#endif
}

int ssg_coll_init(int rank, int world, const char *rdv, ssg_coll_t **out)
{
	*out = 0;
	if (world < 1 || rank < 0 || rank >= world || !rdv) { ssg_err_msg = "ssg_coll_init: bad arguments"; return SSG_EINVAL; }
#ifdef SSG_EMU
	ssg_err_msg = "no RCCL in the host emulation (the ranks' socket transport is the stand-in there)"; return SSG_ENODEV;
#else
	if (rt_device_count() < 1) { ssg_err_msg = "no HIP device visible: libssgpu has no CPU path"; return SSG_ENODEV; }
	if (!rccl.load()) { ssg_err_msg = "librccl.so cannot be loaded"; return SSG_ENODEV; }
	CHK(rt_check(hipSetDevice(ssg_cur_dev), "hipSetDevice"));
	ncclUniqueId id;
	const std::string path = std::string(rdv) + "/rccl.id";
	if (rank == 0) {
		NCHK(rccl.GetUniqueId(&id), "ncclGetUniqueId");
		const std::string tmp = path + ".tmp";
		FILE *f = fopen(tmp.c_str(), "wb");
		if (!f || fwrite(&id, 1, sizeof(id), f) != sizeof(id) || fclose(f) != 0 || rename(tmp.c_str(), path.c_str()) != 0) { ssg_err_msg = "cannot write " + path; return SSG_EIO; }
	} else {
		struct stat sb; double waited = 0;
		while (stat(path.c_str(), &sb) != 0 || (size_t)sb.st_size != sizeof(id)) { if (waited > 600) { ssg_err_msg = path + " did not appear"; return SSG_EIO; } usleep(20000); waited += 0.02; }
		FILE *f = fopen(path.c_str(), "rb");
		if (!f || fread(&id, 1, sizeof(id), f) != sizeof(id)) { if (f) fclose(f); ssg_err_msg = "cannot read " + path; return SSG_EIO; }
		fclose(f);
	}
	ssg_coll *c = new ssg_coll(); c->rank = rank; c->world = world;
	{ ncclResult_t r = rccl.CommInitRank(&c->comm, world, id, rank); if (r != ncclSuccess) { ssg_err_msg = std::string("ncclCommInitRank: ") + rccl.GetErrorString(r); delete c; return SSG_EHIP; } }
	if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); rccl.CommDestroy(c->comm); delete c; ssg_err_msg = "hipStreamCreate failed"; return SSG_EHIP; }
	*out = c;
	return 0;
#endif
}

/* block q of `send` (send_bytes[q] bytes, host) goes to rank q; block q of `recv` receives recv_bytes[q] bytes from rank q -- the sizes are the caller's to
 * agree on beforehand (ssg_coll_alltoall_u64).  Staged through two HBM buffers; one group of world sends and world receives. */
int ssg_coll_alltoallv(ssg_coll_t *c, const void *const *send, const uint64_t *send_bytes, void *const *recv, const uint64_t *recv_bytes)
{
#ifdef SSG_EMU
	(void)c; (void)send; (void)send_bytes; (void)recv; (void)recv_bytes; ssg_err_msg = "no RCCL in the host emulation"; return SSG_ENODEV;
#else
	const int W = c->world;
	uint64_t ts = 0, tr = 0; std::vector<uint64_t> so((size_t)W + 1, 0), ro((size_t)W + 1, 0);
	for (int q = 0; q < W; ++q) { so[(size_t)q + 1] = so[(size_t)q] + ((send_bytes[q] + 255) & ~255ull); ro[(size_t)q + 1] = ro[(size_t)q] + ((recv_bytes[q] + 255) & ~255ull); }
	ts = so[(size_t)W]; tr = ro[(size_t)W];
	uint8_t *ds = (uint8_t*)rt_malloc(ts + 256), *dr = (uint8_t*)rt_malloc(tr + 256);
	if (!ds || !dr) { rt_free(ds); rt_free(dr); ssg_err_msg = "device allocation failed: exchange buffers"; return SSG_ENOMEM; }
	int rc = 0;
	for (int q = 0; q < W && !rc; ++q) if (send_bytes[q]) rc = rt_check(hipMemcpyAsync(ds + so[(size_t)q], send[q], send_bytes[q], hipMemcpyHostToDevice, c->st), "hipMemcpyAsync H2D");
	if (!rc) {
		ncclResult_t r = rccl.GroupStart();
		for (int q = 0; q < W && r == ncclSuccess; ++q) {
			if (send_bytes[q]) r = rccl.Send(ds + so[(size_t)q], send_bytes[q], ncclUint8, q, c->comm, c->st);
			if (r == ncclSuccess && recv_bytes[q]) r = rccl.Recv(dr + ro[(size_t)q], recv_bytes[q], ncclUint8, q, c->comm, c->st);
		}
		const ncclResult_t e = rccl.GroupEnd();
		if (r == ncclSuccess) r = e;
		if (r != ncclSuccess) { ssg_err_msg = std::string("RCCL all-to-all: ") + rccl.GetErrorString(r); rc = SSG_EHIP; }
	}
	for (int q = 0; q < W && !rc; ++q) if (recv_bytes[q]) rc = rt_check(hipMemcpyAsync(recv[q], dr + ro[(size_t)q], recv_bytes[q], hipMemcpyDeviceToHost, c->st), "hipMemcpyAsync D2H");
	if (!rc) rc = rt_check(hipStreamSynchronize(c->st), "hipStreamSynchronize"); else (void)hipStreamSynchronize(c->st);
	rt_free(ds); rt_free(dr);
	return rc;
#endif
}

/* k values to every rank, k from every rank: send[q * k ..], recv[q * k ..] */
int ssg_coll_alltoall_u64(ssg_coll_t *c, const uint64_t *send, uint64_t *recv, int k)
{
	std::vector<const void*> sp((size_t)c->world); std::vector<void*> rp((size_t)c->world); std::vector<uint64_t> n((size_t)c->world, (uint64_t)k * 8);
	for (int q = 0; q < c->world; ++q) { sp[(size_t)q] = send + (size_t)q * k; rp[(size_t)q] = recv + (size_t)q * k; }
	return ssg_coll_alltoallv(c, sp.data(), n.data(), rp.data(), n.data());
}

void ssg_coll_destroy(ssg_coll_t *c)
{
	if (!c) return;
#ifndef SSG_EMU
	if (c->st) (void)hipStreamDestroy(c->st);
	if (c->comm) (void)rccl.CommDestroy(c->comm);
#endif
	delete c;
}

} /* extern "C" */
