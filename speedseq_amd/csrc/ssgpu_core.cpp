/*
 * ssgpu_core.cpp -- host orchestration and C ABI of libssgpu (see include/ssgpu.h).
 * Compiled by hipcc for gfx950 (product) or by g++ with -DSSG_EMU against tests/emu (CPU tests).
 *
 * Stage order for a batch of reads (all intermediates stay in HBM):
 *   ssg_k_smem2 (+ ssg_k_smem_heavy; ssg_seed.cpp) -> ssg_k_smem_sort -> ssg_k_sal_count -> [prefix sum] -> ssg_k_sal -> ssg_k_chain -> ssg_k_chain2aln
 * Per-read variable-length outputs are placed by prefix sums over per-read counts; fixed-capacity
 * stages report overflow and the affected reads are re-run with a larger capacity -- nothing is
 * dropped silently and nothing falls back to the CPU.
 */
#include <vector>
#include <memory>
#include <chrono>
#include <algorithm>
#include <math.h>
#include <atomic>
#include <mutex>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include "ssg_rt.h"
#include "k_seed.h"
#include "k_chainw.h"
#include "k_extend.h"
#include "k_swjobs.h"
#include "k_aln.h"
#include "k_pairw.h"
#include "k_sbl.h"
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"
#include "ssg_pe_int.h"
#ifndef SSG_EMU
#include <rocprim/rocprim.hpp>
#endif

thread_local std::string ssg_err_msg;
thread_local int ssg_cur_dev = 0;
thread_local int ssg_lane = 0;
#ifndef SSG_EMU
ssg_pool_t ssg_pools[SSG_MAX_DEV][SSG_MAX_LANE];
thread_local hipStream_t ssg_stream = 0;
ssg_hostpool_t ssg_hostpool;
int ssg_prof_on = 0;
thread_local std::vector<ssg_prof_rec> ssg_prof_pending;
#endif

/* wave-per-item kernels are grid-strided over at most this many 4-wave workgroups (256 CUs x 4),
 * so per-wave scratch slabs are sized by residency, not by batch size */
#define SSG_MAX_RESIDENT_WG 1024
/* longest read the DP kernels are laid out for (LDS rows, 8-bit columns): SSG_MAX_READ_LEN, ssg_pe_int.h */
#define SSG_STR_(x) #x
#define SSG_STR(x) SSG_STR_(x)

#define CHK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
#define CHKA(b) do { if (!(b).ok()) { ssg_err_msg = "device allocation failed: " #b; return SSG_ENOMEM; } } while (0)

/* the instance of a kernel templated on WIDE (k_sw.h): the one with the fifth register column only for a batch that has a read above 255 bases */
#define SSG_LAUNCH_W(wide, kern, ...) do { if (wide) SSG_LAUNCH(kern<true>, __VA_ARGS__); else SSG_LAUNCH(kern<false>, __VA_ARGS__); } while (0)
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
static int ssg_debug() { static int d = -1; if (d < 0) d = getenv("SSG_DEBUG") ? atoi(getenv("SSG_DEBUG")) : 0; return d; }
static double ssg_stage_ms() { static thread_local std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); const auto t1 = std::chrono::steady_clock::now(); const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count(); t0 = t1; return ms; }
#define STAGE(name) do { if (ssg_debug()) { int rc_ = rt_sync(); fprintf(stderr, "[ssgpu] stage %s done rc=%d  +%.1f ms\n", name, rc_, ssg_stage_ms()); fflush(stderr); if (rc_) return rc_; } } while (0)

SSG_ABI_FP_DEFINE(core)
extern "C" void ssg_abi_fp_index_build(ssg_abi_fp_t*); extern "C" void ssg_abi_fp_seed(ssg_abi_fp_t*); extern "C" void ssg_abi_fp_bgzf(ssg_abi_fp_t*); extern "C" void ssg_abi_fp_sam_format(ssg_abi_fp_t*); extern "C" void ssg_abi_fp_bam(ssg_abi_fp_t*); extern "C" void ssg_abi_fp_coll(ssg_abi_fp_t*);
extern "C" int ssg_abi_selfcheck(void)
{	/* every translation unit of the library was compiled against the same shared declarations (ssg_index_int.h) */
	static const char *const field[20] = { "sizeof(ssg_index_view_t)", "sizeof(ssg_mem_opt_t)", "sizeof(ssg_index)", "sizeof(ssg_intv_t)", "ssg_index_view_t.primary", "ssg_index_view_t.L2",
		"ssg_index_view_t.l_pac", "ssg_index_view_t.sa_intv", "ssg_mem_opt_t.min_seed_len", "ssg_mem_opt_t.split_width", "ssg_mem_opt_t.max_mem_intv", "ssg_mem_opt_t.split_factor", "ssg_mem_opt_t.mat",
		"ssg_index.bwt", "ssg_index.ktab", "ssg_index.bwt_words", "ssg_index.names", "sizeof(ssg_seed_t)", "sizeof(ssg_alnreg_t)", "sizeof(ssg_aln_t)" };
	struct { const char *unit; void (*fn)(ssg_abi_fp_t*); } const units[] = { { "ssg_index_build", ssg_abi_fp_index_build }, { "ssg_seed", ssg_abi_fp_seed }, { "ssg_bgzf", ssg_abi_fp_bgzf }, { "sam_format", ssg_abi_fp_sam_format }, { "ssg_bam", ssg_abi_fp_bam }, { "ssg_coll", ssg_abi_fp_coll } };
	ssg_abi_fp_t mine; ssg_abi_fp_core(&mine);
	for (const auto &u : units) {
		ssg_abi_fp_t o; u.fn(&o);
		for (int i = 0; i < 20; ++i) if (o.v[i] != mine.v[i]) {
			ssg_err_msg = std::string("libssgpu was linked from objects compiled against different declarations: ") + field[i] + " is " + std::to_string(mine.v[i]) + " in ssgpu_core and " + std::to_string(o.v[i]) + " in " + u.unit + " (rebuild: make clean lib)";
			return SSG_EINVAL;
		}
	}
	return 0;
}
static int need_device()
{
	if (rt_device_count() < 1) { ssg_err_msg = "no HIP device visible: libssgpu has no CPU path"; return SSG_ENODEV; }
	return ssg_abi_selfcheck();
}
int ssg_need_device() { return need_device(); }

extern "C" {

const char *ssg_version(void) { return "0.1.0"; }
const char *ssg_backend(void) { return SSG_BACKEND; }
int ssg_device_count(void) { return rt_device_count(); }
int ssg_set_device(int dev) { return rt_set_device(dev); }
int ssg_set_lane(int lane) { return rt_set_lane(lane); }
const char *ssg_last_error(void) { return ssg_err_msg.c_str(); }
void ssg_free(void *p) { free(p); }

void ssg_mem_opt_init(ssg_mem_opt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->a = 1; o->b = 4; o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1;
	o->w = 100; o->T = 30; o->zdrop = 100; o->pen_unpaired = 17; o->pen_clip5 = o->pen_clip3 = 5;
	o->max_mem_intv = 20; o->min_seed_len = 19; o->split_width = 10; o->max_occ = 500;
	o->max_chain_gap = 10000; o->max_ins = 10000; o->mask_level = 0.50f; o->drop_ratio = 0.50f;
	o->XA_drop_ratio = 0.80f; o->split_factor = 1.5f; o->chunk_size = 10000000; o->n_threads = 1;
	o->max_XA_hits = 5; o->max_XA_hits_alt = 200; o->max_matesw = 50; o->mask_level_redun = 0.95f;
	o->min_chain_weight = 0; o->max_chain_extend = 1 << 30;
	o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int)log((double)o->mapQ_coef_len);
	for (int i = 0, k = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) o->mat[k++] = i == j ? o->a : -o->b; o->mat[k++] = -1; }
	for (int j = 0; j < 5; ++j) o->mat[20 + j] = -1;
}

/* ------------------------------- index ------------------------------- */
/* HBM-resident SA sampled more densely than the file (see ssg_k_sa_densify); SSG_SA_INTV overrides the interval (power of two) */
static int densify_sa(ssg_index *ix, int to = 0)
{
	const int want = to > 0 ? to : env_int("SSG_SA_INTV", 4);   /* SAL per million pairs: 8 ms at 4, 67 ms at the file's 32 (MI355X, 60 M seeds); 2 bytes of HBM per reference base at 4 */
	if (want <= 0 || want >= ix->v.sa_intv || (want & (want - 1)) || ix->v.sa_intv % want) return 0;
	const long n_new = (long)((ix->v.seq_len + (uint64_t)want) / (uint64_t)want);
	uint64_t *d = (uint64_t*)rt_malloc((size_t)n_new * 8);
	if (!d) { ssg_err_msg = "index allocation failed: dense SA"; return SSG_ENOMEM; }
	const unsigned long long n_old = (unsigned long long)((ix->v.seq_len + (uint64_t)ix->v.sa_intv) / (uint64_t)ix->v.sa_intv);
	dbuf<unsigned long long> d_next(1);
	if (!d_next.ok()) { rt_free(d); ssg_err_msg = "index allocation failed: dense SA"; return SSG_ENOMEM; }
	CHK(d_next.zero());
	const long n_wg = std::min<long>((long)((n_old + 255) / 256), 256L * env_int("SSG_DENSIFY_WG_PER_CU", 8));   /* persistent: lanes refill from the counter */
	SSG_LAUNCH(ssg_k_sa_densify_walk, n_wg, 256, 0, ix->v, want, d, n_old, d_next.p, std::max(1, std::min(64, env_int("SSG_DENSIFY_REFILL", 32))));
	CHK(rt_sync());
	if (env_int("SSG_SA_VERIFY", 0)) CHK(ssg_sa_verify(ix, want, d, n_new));
	rt_free(ix->sa);   /* the lower-density copy, when this index owns it */
	ix->sa = d; ix->v.sa = d; ix->v.sa_intv = want;
	return 0;
}

int ssg_index_from_arrays(const uint32_t *bwt, uint64_t bwt_words, uint64_t primary, const uint64_t L2[5],
                          const uint64_t *sa, uint64_t n_sa, int sa_intv, const uint8_t *pac, int64_t l_pac,
                          int n_ctg, const int64_t *ctg_off, const int32_t *ctg_len, ssg_index_t **out)
{
	CHK(need_device());
	ssg_index *ix = new ssg_index();
	size_t pac_bytes = (size_t)(l_pac / 4 + 1);
	ix->bwt_words = bwt_words;
	ix->bwt = (uint32_t*)rt_malloc(bwt_words * 4 + 64); ix->sa = (uint64_t*)rt_malloc(n_sa * 8);
	ix->pac = (uint8_t*)rt_malloc(pac_bytes); ix->ctg_off = (int64_t*)rt_malloc(n_ctg * 8); ix->ctg_len = (int32_t*)rt_malloc(n_ctg * 4);
	if (!ix->bwt || !ix->sa || !ix->pac || !ix->ctg_off || !ix->ctg_len) { ssg_index_destroy(ix); ssg_err_msg = "index allocation failed"; return SSG_ENOMEM; }
	int rc = 0;
	rc |= rt_h2d(ix->bwt, bwt, bwt_words * 4); rc |= rt_h2d(ix->sa, sa, n_sa * 8); rc |= rt_h2d(ix->pac, pac, pac_bytes);
	rc |= rt_h2d(ix->ctg_off, ctg_off, n_ctg * 8); rc |= rt_h2d(ix->ctg_len, ctg_len, n_ctg * 4);
	if (rc) { ssg_index_destroy(ix); return SSG_EHIP; }
	ix->v.bwt = ix->bwt; ix->v.sa = ix->sa; ix->v.pac = ix->pac; ix->v.ctg_off = ix->ctg_off; ix->v.ctg_len = ix->ctg_len;
	ix->v.primary = primary; for (int i = 0; i < 5; ++i) ix->v.L2[i] = L2[i];
	ix->v.seq_len = L2[4]; ix->v.l_pac = l_pac; ix->v.n_ctg = n_ctg; ix->v.sa_intv = sa_intv;
	ix->h_off.assign(ctg_off, ctg_off + n_ctg); ix->h_len.assign(ctg_len, ctg_len + n_ctg);
	{ int rc2 = densify_sa(ix); if (!rc2) rc2 = ssg_index_build_ktab(ix); if (rc2) { ssg_index_destroy(ix); return rc2; } }
	*out = ix;
	return 0;
}

/* file bytes -> HBM for the index load: reader threads pread() 16 MB pieces into page-locked staging blocks and send each to the
 * device with an asynchronous copy on the thread's own stream, two blocks per thread, so reading the next piece overlaps the copy
 * of the previous one and nothing is staged in pageable memory (round 2 read the files into zero-filled vectors one after the other
 * and copied them synchronously: 4.6 s of an 8.6 s run at the 3.1 Gbp size) */
struct load_item_t { int fd; uint64_t foff; uint8_t *dst; size_t n; };
static int stream_files_to_device(const std::vector<load_item_t> &items, int n_threads)
{
	const size_t PIECE = (size_t)16 << 20;
	struct piece_t { int fd; uint64_t foff; uint8_t *dst; size_t n; };
	std::vector<piece_t> pieces;
	for (const load_item_t &it : items) for (size_t o = 0; o < it.n; o += PIECE) { piece_t q; q.fd = it.fd; q.foff = it.foff + o; q.dst = it.dst + o; q.n = std::min(PIECE, it.n - o); pieces.push_back(q); }
	if (pieces.empty()) return 0;
	n_threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, pieces.size()));
	std::atomic<size_t> next(0); std::atomic<int> fail(0);
	const int dev = ssg_cur_dev;
	auto pread_all = [](int fd, void *buf, size_t n, uint64_t off) -> bool {
		uint8_t *b = (uint8_t*)buf;
		while (n) { ssize_t r = pread(fd, b, n, (off_t)off); if (r < 0) { if (errno == EINTR) continue; return false; } if (r == 0) return false; b += r; n -= (size_t)r; off += (uint64_t)r; }
		return true;
	};
	auto worker = [&]() {
#ifdef SSG_EMU
		for (;;) { const size_t i = next.fetch_add(1); if (i >= pieces.size() || fail) break; if (!pread_all(pieces[i].fd, pieces[i].dst, pieces[i].n, pieces[i].foff)) fail = 1; }
#else
		if (rt_set_device(dev)) { fail = 2; return; }
		void *stage[2] = { rt_host_alloc(PIECE), rt_host_alloc(PIECE) }; hipStream_t st = 0; hipEvent_t ev[2]; bool busy[2] = { false, false };
		if (!stage[0] || !stage[1] || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { fail = 2; rt_host_free(stage[0]); rt_host_free(stage[1]); return; }
		(void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
		for (int k = 0; ; k ^= 1) {
			const size_t i = next.fetch_add(1);
			if (i >= pieces.size() || fail) break;
			if (busy[k]) { if (hipEventSynchronize(ev[k]) != hipSuccess) { fail = 2; break; } busy[k] = false; }
			if (!pread_all(pieces[i].fd, stage[k], pieces[i].n, pieces[i].foff)) { fail = 1; break; }
			if (hipMemcpyAsync(pieces[i].dst, stage[k], pieces[i].n, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { fail = 2; break; }
			busy[k] = true;
		}
		if (hipStreamSynchronize(st) != hipSuccess) fail = 2;
		(void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]); (void)hipStreamDestroy(st);
		rt_host_free(stage[0]); rt_host_free(stage[1]);
#endif
	};
	std::vector<std::thread> th;
	for (int t = 1; t < n_threads; ++t) th.emplace_back(worker);
	worker();
	for (std::thread &x : th) x.join();
	if (fail == 1) { ssg_err_msg = "short read of an index file"; return SSG_EIO; }
	if (fail) { ssg_err_msg = "index upload failed"; return SSG_EHIP; }
	return 0;
}

int ssg_index_load(const char *prefix, ssg_index_t **out) { return ssg_index_load2(prefix, 0, out); }
int ssg_index_densify(ssg_index_t *ix)
{
	CHK(need_device());
	if (!ix) { ssg_err_msg = "ssg_index_densify: no index"; return SSG_EINVAL; }
	return densify_sa(ix);
}
/* ... to every intv-th row (a power of two below the current interval; anything else is a no-op).  The walk visits every row once whatever the target density:
 * a sparser copy is not cheaper to make (measured: 1.0 s to every 16th row as to every 4th), only smaller. */
int ssg_index_densify_to(ssg_index_t *ix, int intv)
{
	CHK(need_device());
	if (!ix || intv <= 0) { ssg_err_msg = "ssg_index_densify_to: bad arguments"; return SSG_EINVAL; }
	return densify_sa(ix, intv);
}
int ssg_index_load2(const char *prefix, int defer_dense_sa, ssg_index_t **out)
{	/* on-disk layout: SURVEY.md Appendix A (verified against the bundled example index) */
	CHK(need_device());
	const auto t_begin = std::chrono::steady_clock::now();
	std::string p(prefix);
	FILE *fp = fopen((p + ".ann").c_str(), "r");
	if (!fp) { ssg_err_msg = "cannot open " + p + ".ann"; return SSG_EIO; }
	long long l_pac; int n_seqs; unsigned seed;
	if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(fp); ssg_err_msg = "bad .ann header"; return SSG_EIO; }
	std::vector<int64_t> off(n_seqs); std::vector<int32_t> len(n_seqs); std::vector<std::string> names(n_seqs);
	for (int i = 0; i < n_seqs; ++i) {
		unsigned gi; char nm[8192]; long long o; int l, na; int c;
		if (fscanf(fp, "%u%8191s", &gi, nm) != 2) { fclose(fp); ssg_err_msg = "bad .ann record"; return SSG_EIO; }
		while ((c = fgetc(fp)) != '\n' && c != EOF) {}
		if (fscanf(fp, "%lld%d%d", &o, &l, &na) != 3) { fclose(fp); ssg_err_msg = "bad .ann record"; return SSG_EIO; }
		names[i] = nm; off[i] = o; len[i] = l;
	}
	fclose(fp);
	struct fd_t { int fd; uint64_t size; fd_t() : fd(-1), size(0) {} ~fd_t() { if (fd >= 0) close(fd); } };
	fd_t f_bwt, f_sa, f_pac;
	auto open_ro = [&](const std::string &fn, fd_t &f) -> int {
		f.fd = open(fn.c_str(), O_RDONLY); struct stat sb;
		if (f.fd < 0 || fstat(f.fd, &sb) != 0) { ssg_err_msg = "cannot open " + fn; return SSG_EIO; }
		f.size = (uint64_t)sb.st_size; return 0;
	};
	CHK(open_ro(p + ".bwt", f_bwt)); CHK(open_ro(p + ".sa", f_sa)); CHK(open_ro(p + ".pac", f_pac));
	if (f_bwt.size < 40 || f_sa.size < 56) { ssg_err_msg = "truncated index"; return SSG_EIO; }
	uint64_t primary, L2[5] = {0,0,0,0,0}, hb[5], hdr[7];
	if (pread(f_bwt.fd, hb, 40, 0) != 40 || pread(f_sa.fd, hdr, 56, 0) != 56) { ssg_err_msg = "cannot read the index headers"; return SSG_EIO; }
	primary = hb[0]; memcpy(L2 + 1, hb + 1, 32);
	if (hdr[0] != primary || hdr[6] != L2[4]) { ssg_err_msg = ".sa does not match .bwt"; return SSG_EIO; }
	const int sa_intv = (int)hdr[5];
	if (sa_intv <= 0) { ssg_err_msg = "bad .sa interval"; return SSG_EIO; }
	const uint64_t n_sa = (L2[4] + sa_intv) / sa_intv;
	if (f_sa.size != 56 + (n_sa - 1) * 8) { ssg_err_msg = "unexpected .sa size"; return SSG_EIO; }
	const size_t pac_bytes = (size_t)(l_pac / 4 + 1), bwt_bytes = (size_t)(f_bwt.size - 40);
	ssg_index *ix = new ssg_index();
	ix->bwt_words = bwt_bytes / 4;
	ix->bwt = (uint32_t*)rt_malloc(bwt_bytes + 64); ix->sa = (uint64_t*)rt_malloc(n_sa * 8);
	ix->pac = (uint8_t*)rt_malloc(pac_bytes); ix->ctg_off = (int64_t*)rt_malloc((size_t)n_seqs * 8); ix->ctg_len = (int32_t*)rt_malloc((size_t)n_seqs * 4);
	if (!ix->bwt || !ix->sa || !ix->pac || !ix->ctg_off || !ix->ctg_len) { ssg_index_destroy(ix); ssg_err_msg = "index allocation failed"; return SSG_ENOMEM; }
	const size_t pac_have = (size_t)std::min<uint64_t>(f_pac.size, pac_bytes);
	int rc = 0;
	if (pac_have < pac_bytes) rc |= rt_memset(ix->pac + pac_have, 0, pac_bytes - pac_have);
	{ const uint64_t none = (uint64_t)-1; rc |= rt_h2d(ix->sa, &none, 8); }                 /* row 0 (the terminator) is not stored in the file */
	rc |= rt_h2d(ix->ctg_off, off.data(), (size_t)n_seqs * 8); rc |= rt_h2d(ix->ctg_len, len.data(), (size_t)n_seqs * 4);
	if (rc) { ssg_index_destroy(ix); return SSG_EHIP; }
	std::vector<load_item_t> items(3);
	items[0].fd = f_bwt.fd; items[0].foff = 40; items[0].dst = (uint8_t*)ix->bwt; items[0].n = bwt_bytes;
	items[1].fd = f_sa.fd; items[1].foff = 56; items[1].dst = (uint8_t*)(ix->sa + 1); items[1].n = (size_t)(n_sa - 1) * 8;
	items[2].fd = f_pac.fd; items[2].foff = 0; items[2].dst = ix->pac; items[2].n = pac_have;
	{ const int rc2 = stream_files_to_device(items, std::max(1, env_int("SSG_LOAD_THREADS", 8))); if (rc2) { ssg_index_destroy(ix); return rc2; } }
	const auto t_up = std::chrono::steady_clock::now();
	ix->v.bwt = ix->bwt; ix->v.sa = ix->sa; ix->v.pac = ix->pac; ix->v.ctg_off = ix->ctg_off; ix->v.ctg_len = ix->ctg_len;
	ix->v.primary = primary; for (int i = 0; i < 5; ++i) ix->v.L2[i] = L2[i];
	ix->v.seq_len = L2[4]; ix->v.l_pac = l_pac; ix->v.n_ctg = n_seqs; ix->v.sa_intv = sa_intv;
	ix->h_off = off; ix->h_len = len; ix->names = names;
	{ int rc2 = defer_dense_sa ? 0 : densify_sa(ix); if (!rc2) rc2 = ssg_index_build_ktab(ix); if (rc2) { ssg_index_destroy(ix); return rc2; } }
	if (ssg_debug() || getenv("SSG_LOAD_LOG")) {
		const auto t_end = std::chrono::steady_clock::now();
		fprintf(stderr, "[ssgpu] index load: files -> HBM %.3f s (%.2f GB), SA samples to every %d rows %.3f s\n", std::chrono::duration<double>(t_up - t_begin).count(),
		        (double)(bwt_bytes + n_sa * 8 + pac_have) / 1e9, ix->v.sa_intv, std::chrono::duration<double>(t_end - t_up).count());
	}
	*out = ix;
	return 0;
}

void ssg_index_destroy(ssg_index_t *ix)
{
	if (!ix) return;
	if (ix->raw_alloc) { rt_free_raw(ix->bwt); rt_free_raw(ix->sa); rt_free_raw(ix->pac); }
	else { rt_free(ix->bwt); rt_free(ix->sa); rt_free(ix->pac); }
	rt_free(ix->ctg_off); rt_free(ix->ctg_len); rt_free(ix->ktab); rt_free(ix->d_names); rt_free(ix->d_name_off);
	delete ix;
}
int ssg_index_from_device(const uint32_t *d_bwt, uint64_t primary, const uint64_t L2[5], const uint64_t *d_sa, int sa_intv,
                          const uint8_t *d_pac, int64_t l_pac, int n_ctg, const int64_t *ctg_off, const int32_t *ctg_len, ssg_index_t **out)
{
	CHK(need_device());
	ssg_index *ix = new ssg_index();
	ix->bwt = 0; ix->sa = 0; ix->pac = 0; /* borrowed: not freed by ssg_index_destroy */
	ix->ctg_off = (int64_t*)rt_malloc(n_ctg * 8); ix->ctg_len = (int32_t*)rt_malloc(n_ctg * 4);
	if (!ix->ctg_off || !ix->ctg_len) { ssg_index_destroy(ix); ssg_err_msg = "index allocation failed"; return SSG_ENOMEM; }
	if (rt_h2d(ix->ctg_off, ctg_off, n_ctg * 8) | rt_h2d(ix->ctg_len, ctg_len, n_ctg * 4)) { ssg_index_destroy(ix); return SSG_EHIP; }
	ix->v.bwt = d_bwt; ix->v.sa = d_sa; ix->v.pac = d_pac; ix->v.ctg_off = ix->ctg_off; ix->v.ctg_len = ix->ctg_len;
	ix->v.primary = primary; for (int i = 0; i < 5; ++i) ix->v.L2[i] = L2[i];
	ix->v.seq_len = L2[4]; ix->v.l_pac = l_pac; ix->v.n_ctg = n_ctg; ix->v.sa_intv = sa_intv;
	ix->h_off.assign(ctg_off, ctg_off + n_ctg); ix->h_len.assign(ctg_len, ctg_len + n_ctg);
	{ int rc2 = densify_sa(ix); if (!rc2) rc2 = ssg_index_build_ktab(ix); if (rc2) { ssg_index_destroy(ix); return rc2; } }
	*out = ix;
	return 0;
}

/* ---- per-kernel timing ---- */
#ifndef SSG_EMU
} /* extern "C" */
static void prof_collect();
void ssg_prof_flush() { prof_collect(); }
extern "C" {
static std::vector<std::string> prof_names; static std::vector<double> prof_ms; static std::vector<long> prof_cnt; static std::mutex prof_mu;
static void prof_collect()
{	/* the calling thread's launches (on its current device) into the process-wide table */
	if (ssg_prof_pending.empty()) return;
	(void)hipDeviceSynchronize();
	std::lock_guard<std::mutex> lk(prof_mu);
	for (auto &r : ssg_prof_pending) {
		float ms = 0; (void)hipEventElapsedTime(&ms, r.a, r.b);
		std::string nm(r.name);
		if (nm.size() > 2 && nm.front() == '(' && nm.back() == ')') nm = nm.substr(1, nm.size() - 2);   /* template instances with two arguments are launched as (kernel<A, B>) */
		size_t i = 0; for (; i < prof_names.size(); ++i) if (prof_names[i] == nm) break;
		if (i == prof_names.size()) { prof_names.push_back(nm); prof_ms.push_back(0); prof_cnt.push_back(0); }
		prof_ms[i] += ms; prof_cnt[i] += 1;
		(void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
	}
	ssg_prof_pending.clear();
}
void ssg_prof_enable(int on) { prof_collect(); ssg_prof_on = on; }
void ssg_prof_reset(void) { prof_collect(); std::lock_guard<std::mutex> lk(prof_mu); prof_names.clear(); prof_ms.clear(); prof_cnt.clear(); }
int ssg_prof_get(int cap, const char **name, double *ms, long *launches)
{
	prof_collect();
	std::lock_guard<std::mutex> lk(prof_mu);
	for (int i = 0; i < (int)prof_names.size() && i < cap; ++i) { name[i] = prof_names[i].c_str(); ms[i] = prof_ms[i]; launches[i] = prof_cnt[i]; }
	return (int)prof_names.size();
}
#else
void ssg_prof_enable(int) {}
void ssg_prof_reset(void) {}
int ssg_prof_get(int, const char **, double *, long *) { return 0; }
#endif

/* tuning aid: device phase counters (cycles) accumulated by instrumented kernels; reset on read */
int ssg_dbg_cycles(unsigned long long out[32])
{
#ifdef SSG_EMU
	memcpy(out, ssg_dbg_cyc, 256); memset(ssg_dbg_cyc, 0, 256);
	return 0;
#else
	unsigned long long z[32]; memset(z, 0, sizeof z);
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ssg_dbg_cyc), 256) != hipSuccess) return SSG_EHIP;
	if (hipMemcpyToSymbol(HIP_SYMBOL(ssg_dbg_cyc), z, 256) != hipSuccess) return SSG_EHIP;
	return 0;
#endif
}
/* any stretch of the counters (tuning builds) */
int ssg_dbg_cycles_at(int first, int n, unsigned long long *out)
{
	if (first < 0 || n < 0 || first + n > 96) return SSG_EINVAL;
#ifdef SSG_EMU
	memcpy(out, ssg_dbg_cyc + first, (size_t)n * 8); memset(ssg_dbg_cyc + first, 0, (size_t)n * 8);
	return 0;
#else
	unsigned long long z[96]; memset(z, 0, sizeof z);
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ssg_dbg_cyc), (size_t)n * 8, (size_t)first * 8) != hipSuccess) return SSG_EHIP;
	if (hipMemcpyToSymbol(HIP_SYMBOL(ssg_dbg_cyc), z, (size_t)n * 8, (size_t)first * 8) != hipSuccess) return SSG_EHIP;
	return 0;
#endif
}
/* test hook: how the last call of this thread split its CIGAR requests (wave kernel, lane DP classes 0..2) */
void ssg_dbg_reg2aln_counts(unsigned int out[4]);
/* slots 32..63 (32..47: the lane-per-extension kernel's lane utilisation, k_extlane.h; 48..63: band widths of the requests that need a DP, k_aln.h) */
int ssg_dbg_cycles_hi(unsigned long long out[32])
{
#ifdef SSG_EMU
	memcpy(out, ssg_dbg_cyc + 32, 256); memset(ssg_dbg_cyc + 32, 0, 256);
	return 0;
#else
	unsigned long long z[32]; memset(z, 0, sizeof z);
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ssg_dbg_cyc), 256, 256) != hipSuccess) return SSG_EHIP;
	if (hipMemcpyToSymbol(HIP_SYMBOL(ssg_dbg_cyc), z, 256, 256) != hipSuccess) return SSG_EHIP;
	return 0;
#endif
}

int64_t ssg_index_l_pac(const ssg_index_t *ix) { return ix->v.l_pac; }
int ssg_index_n_ctg(const ssg_index_t *ix) { return ix->v.n_ctg; }

/* ------------------------------- SW job batches ------------------------------- */
int ssg_extend_batch(const ssg_mem_opt_t *opt, int n_jobs, const ssg_ext_job_t *jobs, const uint8_t *qbuf, size_t qbytes,
                     const uint8_t *tbuf, size_t tbytes, ssg_ext_res_t *res, uint64_t *cells)
{
	CHK(need_device());
	if (n_jobs <= 0) return 0;
	for (int i = 0; i < n_jobs; ++i) if (jobs[i].qlen > 318 || jobs[i].qlen < 0 || jobs[i].h0 <= 0) { ssg_err_msg = "ssg_extend_batch: qlen must be <= 318 and h0 > 0"; return SSG_EINVAL; }
	dbuf<ssg_ext_job_t> dj(n_jobs); dbuf<uint8_t> dq(qbytes + 1), dt(tbytes + 1); dbuf<ssg_ext_res_t> dr(n_jobs); dbuf<unsigned long long> dc(1);
	CHKA(dj); CHKA(dq); CHKA(dt); CHKA(dr); CHKA(dc);
	CHK(dj.up(jobs, n_jobs)); CHK(dq.up(qbuf, qbytes)); CHK(dt.up(tbuf, tbytes)); CHK(dc.zero());
	int wpb = 4;
	SSG_LAUNCH(ssg_k_extend_jobs, (n_jobs + wpb - 1) / wpb, wpb * 64, 0, *opt, n_jobs, dj.p, dq.p, dt.p, dr.p, dc.p);
	CHK(rt_sync());
	CHK(dr.down(res, n_jobs));
	if (cells) { unsigned long long c; CHK(dc.down(&c, 1)); *cells = c; }
	return 0;
}

int ssg_extend_lane_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_jobs, const ssg_ext_job_t *jobs, const int64_t *tpos, int dir,
                          const uint8_t *qbuf, size_t qbytes, int qcap, ssg_ext_res_t *res, uint64_t *cells)
{
	CHK(need_device());
	if (n_jobs <= 0) return 0;
	if (qcap != 72 && qcap != 136 && qcap != 256 && qcap != 320) { ssg_err_msg = "ssg_extend_lane_batch: qcap is 72, 136, 256 or 320"; return SSG_EINVAL; }
	if (dir != 1 && dir != -1) { ssg_err_msg = "ssg_extend_lane_batch: dir is +1 or -1"; return SSG_EINVAL; }
	for (int i = 0; i < n_jobs; ++i) {
		if (jobs[i].qlen > std::min(qcap, 318) || jobs[i].qlen < 0 || jobs[i].h0 <= 0 || jobs[i].tlen < 0) { ssg_err_msg = "ssg_extend_lane_batch: 0 <= qlen <= min(qcap, 318), h0 > 0, tlen >= 0"; return SSG_EINVAL; }
		if ((long)jobs[i].h0 + (long)jobs[i].qlen * opt->a + std::max(jobs[i].end_bonus, 0) >= 8191) { ssg_err_msg = "ssg_extend_lane_batch: h0 + qlen * a exceeds the 13-bit DP cells"; return SSG_EINVAL; }
		const int64_t last = tpos[i] + (int64_t)dir * (jobs[i].tlen - 1);
		if (jobs[i].tlen > 0 && (tpos[i] < 0 || tpos[i] >= 2 * idx->v.l_pac || last < 0 || last >= 2 * idx->v.l_pac || (tpos[i] < idx->v.l_pac) != (last < idx->v.l_pac))) {
			ssg_err_msg = "ssg_extend_lane_batch: a target leaves its strand of the doubled reference"; return SSG_EINVAL; }
	}
	dbuf<ssg_ext_job_t> dj(n_jobs); dbuf<int64_t> dp(n_jobs); dbuf<uint8_t> dq(qbytes + 1); dbuf<ssg_ext_res_t> dr(n_jobs); dbuf<unsigned long long> dc(1);
	CHKA(dj); CHKA(dp); CHKA(dq); CHKA(dr); CHKA(dc);
	CHK(dj.up(jobs, n_jobs)); CHK(dp.up(tpos, n_jobs)); CHK(dq.up(qbuf, qbytes)); CHK(dc.zero()); CHK(dr.zero());
	const long nb = (n_jobs + 63) / 64;
	if (qcap == 72) SSG_LAUNCH(ssg_k_ext_lane_jobs<72>, nb, 64, 0, idx->v, *opt, n_jobs, dj.p, dp.p, dir, dq.p, dr.p, dc.p);
	else if (qcap == 136) SSG_LAUNCH(ssg_k_ext_lane_jobs<136>, nb, 64, 0, idx->v, *opt, n_jobs, dj.p, dp.p, dir, dq.p, dr.p, dc.p);
	else if (qcap == 256) SSG_LAUNCH(ssg_k_ext_lane_jobs<256>, nb, 64, 0, idx->v, *opt, n_jobs, dj.p, dp.p, dir, dq.p, dr.p, dc.p);
	else SSG_LAUNCH(ssg_k_ext_lane_jobs<320>, nb, 64, 0, idx->v, *opt, n_jobs, dj.p, dp.p, dir, dq.p, dr.p, dc.p);
	CHK(rt_sync());
	CHK(dr.down(res, n_jobs));
	if (cells) { unsigned long long c; CHK(dc.down(&c, 1)); *cells = c; }
	return 0;
}

int ssg_align2_batch(const ssg_mem_opt_t *opt, int n_jobs, const ssg_sw_job_t *jobs, const uint8_t *qbuf, size_t qbytes,
                     const uint8_t *tbuf, size_t tbytes, ssg_kswr_t *res)
{
	CHK(need_device());
	if (n_jobs <= 0) return 0;
	int max_t = 1;
	for (int i = 0; i < n_jobs; ++i) { if (jobs[i].qlen > 320 || jobs[i].qlen < 1) { ssg_err_msg = "ssg_align2_batch: 1 <= qlen <= 320"; return SSG_EINVAL; } max_t = std::max(max_t, jobs[i].tlen); }
	dbuf<ssg_sw_job_t> dj(n_jobs); dbuf<uint8_t> dq(qbytes + 1), dt(tbytes + 1); dbuf<ssg_kswr_t> dr(n_jobs); dbuf<unsigned long long> db((size_t)n_jobs * (max_t + 1));
	CHKA(dj); CHKA(dq); CHKA(dt); CHKA(dr); CHKA(db);
	CHK(dj.up(jobs, n_jobs)); CHK(dq.up(qbuf, qbytes)); CHK(dt.up(tbuf, tbytes));
	int wpb = 4;
	SSG_LAUNCH(ssg_k_align2_jobs, (n_jobs + wpb - 1) / wpb, wpb * 64, 0, *opt, n_jobs, dj.p, dq.p, dt.p, dr.p, db.p, max_t + 1);
	CHK(rt_sync());
	return dr.down(res, n_jobs);
}

static int sort_keys_u64(uint64_t *k_in, uint64_t *k_out, long n, int begin_bit, int end_bit);
/* The forward passes of nj mate-rescue windows (k_mswlane.h): d_keys[nj] = qp << 48 | tlen << 32 | slot (unsorted; sorted here so that a
 * wavefront's jobs share the padded query length and have similar window lengths), results into d_res[slot].  lanes = 1, 2 or 4 per job;
 * max_qp / max_tlen = the largest padded query length / window length among the jobs (LDS columns per lane = max_qp / lanes;
 * b[] entries per job = max_tlen / 2 + 2). */
static int run_msw_lane(const ssg_index_t *idx, const ssg_mem_opt_t *opt, long nj, uint64_t *d_keys, const ssg_msjob_t *d_jobs, const uint8_t *d_seq, ssg_msres_t *d_res,
                        int max_qp, int max_tlen, int lanes, unsigned long long *d_cells, long n_slots, long seq_bytes, bool rev = false)
{
	if (nj <= 0) return 0;
	if (lanes != 1 && lanes != 2 && lanes != 4) { ssg_err_msg = "mate rescue lane kernel: 1, 2 or 4 lanes per job"; return SSG_EINVAL; }
	dbuf<uint64_t> d_sorted((size_t)nj); dbuf<unsigned int> d_q(1);
	CHKA(d_sorted); CHKA(d_q); CHK(d_q.zero());
	/* bits [32, 58): padded query length (<= 256 at bit 48) over window length (< 2^16).  NOT [32, 64): the radix sort of ROCm 7.2 (met through hipCUB, the same code underneath) returns
	 * a list that is not even a permutation of its input for that range of 64-bit keys (tools/dbg/sort_probe.cpp) */
	CHK(sort_keys_u64(d_keys, d_sorted.p, nj, 32, 58));
	STAGE("msw_sort");
	if (ssg_debug() >= 2) {	/* diagnostic: the sorted list is a permutation of the keys, in order, and every key names a plausible job */
		std::vector<uint64_t> hk((size_t)nj), hs((size_t)nj);
		CHK(rt_d2h(hk.data(), d_keys, (size_t)nj * 8)); CHK(rt_d2h(hs.data(), d_sorted.p, (size_t)nj * 8));
		long bad_slot = 0, bad_order = 0, bad_job = 0;
		std::vector<ssg_msjob_t> hj((size_t)n_slots);
		CHK(rt_d2h(hj.data(), d_jobs, (size_t)n_slots * sizeof(ssg_msjob_t)));
		for (long i = 0; i < nj; ++i) {
			const uint64_t sl = hk[(size_t)i] & 0xffffffffu;
			if ((long)sl >= n_slots) { ++bad_slot; continue; }
			const ssg_msjob_t &jb = hj[(size_t)sl];
			if (!(jb.qlen >= 1 && jb.qlen <= 320 && jb.tlen >= 1 && jb.tlen <= SSG_ML_TMAX && (uint64_t)jb.qp == hk[(size_t)i] >> 48 && (uint64_t)jb.tlen == (hk[(size_t)i] >> 32 & 0xffff))) ++bad_job;
		}
		for (long i = 1; i < nj; ++i) if ((hs[(size_t)i] >> 32) < (hs[(size_t)i - 1] >> 32)) ++bad_order;
		std::sort(hk.begin(), hk.end()); std::vector<uint64_t> hs2(hs); std::sort(hs2.begin(), hs2.end());
		fprintf(stderr, "[ssgpu] msw keys: %ld keys, %ld with a slot out of range, %ld naming a job that does not match, %ld out of order after the sort, same multiset: %d\n", nj, bad_slot, bad_job, bad_order, (int)(hk == hs2));
	}
	const int J = 64 / lanes, ccap = (max_qp + lanes - 1) / lanes + 1;
	const size_t lds = (size_t)(ccap + 2) * 64 * 4;
	const long nchunk = (nj + J - 1) / J;
	const long resident = 256L * std::min<long>(16, (long)(160 * 1024 / lds));
	const long grid = std::min(nchunk, std::max(resident, 1L));
	const int bcap = rev ? 1 : max_tlen / 2 + 2;
	dbuf<unsigned long long> d_bl((size_t)grid * J * bcap);
	CHKA(d_bl);
#define SSG_ML_GO(LL, RR) SSG_LAUNCH((ssg_k_msw_lane<LL, RR>), grid, 64, lds, idx->v, *opt, nj, d_sorted.p, d_jobs, d_seq, d_res, d_bl.p, d_q.p, ccap, bcap, d_cells, n_slots, seq_bytes)
	if (rev) { if (lanes == 4) SSG_ML_GO(4, true); else if (lanes == 2) SSG_ML_GO(2, true); else SSG_ML_GO(1, true); }
	else { if (lanes == 4) SSG_ML_GO(4, false); else if (lanes == 2) SSG_ML_GO(2, false); else SSG_ML_GO(1, false); }
#undef SSG_ML_GO
	if (!rev && env_int("SSG_MSW_REV", 1)) {	/* the reverse passes (KSW_XSTART) of the windows that call for one, the same way */
		dbuf<uint64_t> d_rkeys((size_t)nj); dbuf<unsigned int> d_nr(2);
		CHKA(d_rkeys); CHKA(d_nr); CHK(d_nr.zero());
		SSG_LAUNCH(ssg_k_msw_revlist, (nj + 63) / 64, 64, 0, nj, d_sorted.p, d_jobs, d_res, n_slots, d_rkeys.p, d_nr.p);
		unsigned int nr[2] = { 0, 0 };
		CHK(d_nr.down(nr, 2));
		if (nr[0]) CHK(run_msw_lane(idx, opt, (long)nr[0], d_rkeys.p, d_jobs, d_seq, d_res, max_qp, (int)nr[1], lanes, 0, n_slots, seq_bytes, true));
	}
	int rc = rt_sync();   /* the temporaries are released on return */
#if defined(SSG_ML_CHECK) && !defined(SSG_EMU)
	{ unsigned long long dc[32]; if (hipMemcpyFromSymbol(dc, HIP_SYMBOL(ssg_dbg_cyc), 256) == hipSuccess) fprintf(stderr, "[ssgpu] lane kernel checks: slot %llu job %llu blist %llu (grid %ld J %d bcap %d ccap %d lds %zu nj %ld)\n", dc[24], dc[25], dc[26], grid, J, bcap, ccap, lds, nj); }
#endif
	return rc;
}

/* test hook: sorts n <= 5120 words (w << 32 | id) by w, descending, as the chaining kernels do -- by the whole wave (out_wave) and by one lane running
 * upstream's introsort (out_lane); the two must be equal word for word */
int ssg_dbg_chain_sort(const int64_t *keys, int n, int64_t *out_lane, int64_t *out_wave)
{
	CHK(need_device());
	if (n < 0 || n > 5120) { ssg_err_msg = "ssg_dbg_chain_sort: 0 <= n <= 5120"; return SSG_EINVAL; }
	dbuf<int64_t> d_in((size_t)n + 1), d_o0((size_t)n + 1), d_o1((size_t)n + 1);
	CHKA(d_in); CHKA(d_o0); CHKA(d_o1);
	CHK(d_in.up(keys, n));
	if (n <= 1024) { SSG_LAUNCH(ssg_k_dbg_chain_sort<1024>, 1, 64, 0, d_in.p, n, 0, d_o0.p); SSG_LAUNCH(ssg_k_dbg_chain_sort<1024>, 1, 64, 0, d_in.p, n, 1, d_o1.p); }
	else { SSG_LAUNCH(ssg_k_dbg_chain_sort<5120>, 1, 64, 0, d_in.p, n, 0, d_o0.p); SSG_LAUNCH(ssg_k_dbg_chain_sort<5120>, 1, 64, 0, d_in.p, n, 1, d_o1.p); }
	CHK(rt_sync());
	CHK(d_o0.down(out_lane, n)); CHK(d_o1.down(out_wave, n));
	return SSG_OK;
}

int ssg_align2_lane_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_jobs, const ssg_sw_job_t *jobs, const int64_t *tpos,
                          const uint8_t *qbuf, size_t qbytes, int lanes, ssg_kswr_t *res, int32_t *from_lane)
{
	CHK(need_device());
	if (n_jobs <= 0) return 0;
	int max_t = 1, max_qp = 16;
	std::vector<ssg_msjob_t> hj((size_t)n_jobs); std::vector<uint64_t> hk;
	for (int i = 0; i < n_jobs; ++i) {
		const ssg_sw_job_t &jb = jobs[i];
		if (jb.qlen > 320 || jb.qlen < 1 || jb.tlen < 0) { ssg_err_msg = "ssg_align2_lane_batch: 1 <= qlen <= 320, tlen >= 0"; return SSG_EINVAL; }
		const int64_t last = tpos[i] + jb.tlen - 1;
		if (jb.tlen > 0 && (tpos[i] < 0 || last >= 2 * idx->v.l_pac || (tpos[i] < idx->v.l_pac) != (last < idx->v.l_pac))) {
			ssg_err_msg = "ssg_align2_lane_batch: a target leaves its strand of the doubled reference"; return SSG_EINVAL; }
		max_t = std::max(max_t, jb.tlen);
		ssg_msjob_t m; m.rb = tpos[i]; m.qoff = jb.qoff; m.tlen = jb.tlen; m.qlen = jb.qlen; m._pad = 0; m.is_rev = 0; m.p = (jb.xtra & SSG_KSW_XBYTE) ? 16 : 8; m.xstart = (jb.xtra & SSG_KSW_XSTART) ? 1 : 0;
		m.qp = (jb.xtra & SSG_KSW_XBYTE) ? (jb.qlen + 15) / 16 * 16 : (jb.qlen + 7) / 8 * 8;
		m.minsc = (jb.xtra & SSG_KSW_XSUBO) ? jb.xtra & 0xffff : 0x10000;
		hj[(size_t)i] = m;
		const bool fits = opt->a >= 1 && opt->a <= 15 && opt->b >= 0 && opt->b <= 16 && jb.qlen * opt->a <= 8190;
		if (fits && !(jb.xtra & SSG_KSW_XSTOP) && jb.tlen > 0 && jb.tlen <= SSG_ML_TMAX) { hk.push_back((uint64_t)m.qp << 48 | (uint64_t)m.tlen << 32 | (uint64_t)i); max_qp = std::max(max_qp, m.qp); }
	}
	dbuf<ssg_sw_job_t> dj(n_jobs); dbuf<ssg_msjob_t> dm(n_jobs); dbuf<ssg_msres_t> df(n_jobs); dbuf<int64_t> dp(n_jobs); dbuf<uint8_t> dq(qbytes + 1), dt((size_t)n_jobs * (max_t + 1));
	dbuf<ssg_kswr_t> dr(n_jobs); dbuf<unsigned long long> db((size_t)n_jobs * (max_t + 1)); dbuf<uint64_t> dk(hk.size() + 1); dbuf<int32_t> du(n_jobs);
	CHKA(dj); CHKA(dm); CHKA(df); CHKA(dp); CHKA(dq); CHKA(dt); CHKA(dr); CHKA(db); CHKA(dk); CHKA(du);
	CHK(dj.up(jobs, n_jobs)); CHK(dm.up(hj.data(), n_jobs)); CHK(dp.up(tpos, n_jobs)); CHK(dq.up(qbuf, qbytes)); CHK(df.zero());
	if (!hk.empty()) { CHK(dk.up(hk.data(), hk.size())); CHK(run_msw_lane(idx, opt, (long)hk.size(), dk.p, dm.p, dq.p, df.p, max_qp, max_t, lanes, 0, (long)n_jobs, (long)qbytes)); }
	const int wpb = 4;
	SSG_LAUNCH(ssg_k_align2_fin_jobs, (n_jobs + wpb - 1) / wpb, wpb * 64, 0, idx->v, *opt, n_jobs, dj.p, dp.p, dq.p, df.p, dr.p, dt.p, max_t + 1, db.p, max_t + 1, du.p);
	CHK(rt_sync());
	if (from_lane) CHK(du.down(from_lane, n_jobs));
	return dr.down(res, n_jobs);
}

int ssg_global_batch(const ssg_mem_opt_t *opt, int n_jobs, const ssg_glb_job_t *jobs, const uint8_t *qbuf, size_t qbytes,
                     const uint8_t *tbuf, size_t tbytes, int32_t *score, int32_t *n_cigar, uint32_t *cigar, int cap)
{
	CHK(need_device());
	if (n_jobs <= 0) return 0;
	long zmax = 1;
	for (int i = 0; i < n_jobs; ++i) {
		if (jobs[i].qlen > 318 || jobs[i].qlen < 1) { ssg_err_msg = "ssg_global_batch: 1 <= qlen <= 318"; return SSG_EINVAL; }
		long ncol = std::min(jobs[i].qlen, 2 * jobs[i].w + 1); zmax = std::max(zmax, ncol * jobs[i].tlen);
	}
	dbuf<ssg_glb_job_t> dj(n_jobs); dbuf<uint8_t> dq(qbytes + 1), dt(tbytes + 1), dz((size_t)zmax * n_jobs); dbuf<int32_t> ds(n_jobs), dn(n_jobs); dbuf<uint32_t> dcg((size_t)n_jobs * cap);
	CHKA(dj); CHKA(dq); CHKA(dt); CHKA(dz); CHKA(ds); CHKA(dn); CHKA(dcg);
	CHK(dj.up(jobs, n_jobs)); CHK(dq.up(qbuf, qbytes)); CHK(dt.up(tbuf, tbytes));
	int wpb = 4;
	SSG_LAUNCH(ssg_k_global_jobs, (n_jobs + wpb - 1) / wpb, wpb * 64, 0, *opt, n_jobs, dj.p, dq.p, dt.p, ds.p, dn.p, dcg.p, cap, dz.p, zmax);
	CHK(rt_sync());
	CHK(ds.down(score, n_jobs)); CHK(dn.down(n_cigar, n_jobs));
	return dcg.down(cigar, (size_t)n_jobs * cap);
}

/* ------------------------------- seeding ------------------------------- */
struct seed_stage_t {
	int cap; dbuf<ssg_intv_t> intv; dbuf<int32_t> n_intv;
};

/* runs the SMEM kernel for all reads (cap0 per read), then re-runs overflowing reads with a
 * private large capacity and copies their lists back; on return every n_intv[r] >= 0. */
static int dev_class_counts(const int32_t *d_key, long n, int tA, int tB, int tC, unsigned int out[5]);
static int sort_pairs_u64(uint64_t *k_in, uint64_t *k_out, uint32_t *v_in, uint32_t *v_out, long n);
/* upstream's ks_introsort(mem_intv) of every read's list (k_seed.h): a lane per read ranks the lists of up to 24 intervals, a wave per read the longer ones;
 * SSG_SMEM_SORT_RANK=0: introsort by a lane per read for all (until r06V) */
static int launch_smem_sort(int n_reads, ssg_intv_t *d_intv, const int32_t *d_n, int cap)
{
	if (env_int("SSG_SMEM_SORT_RANK", 1) == 0) { SSG_LAUNCH(ssg_k_smem_sort, (n_reads + 63) / 64, 64, 0, n_reads, d_intv, d_n, cap); return 0; }
	dbuf<int32_t> d_todo((size_t)n_reads); dbuf<unsigned int> d_nt(1);
	if (!d_todo.ok() || !d_nt.ok()) { ssg_err_msg = "device allocation failed: interval sort work list"; return SSG_ENOMEM; }
	CHK(d_nt.zero());
	SSG_LAUNCH(ssg_k_smem_sort_rank<24>, (n_reads + 63) / 64, 64, 0, n_reads, d_intv, d_n, cap, d_todo.p, d_nt.p);
	SSG_LAUNCH(ssg_k_smem_sort_wave, std::min(n_reads, 256 * 32), 64, 0, d_intv, d_n, cap, (const int32_t*)d_todo.p, (const unsigned int*)d_nt.p);
	return rt_sync();   /* (before the work list goes back to the arena) */
}

static int run_smem(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                    int max_len, int cap, ssg_intv_t *d_intv, int32_t *d_n, unsigned long long *n_extend = 0, int *need_cap = 0)
{
	const int block = 64;
	/* SSG_SMEM_KERNEL=lane: the nested-loop form (k_seed.h) instead of the product's kernels (ssg_seed.cpp: ssg_k_smem2 + ssg_k_smem_heavy) */
	const bool smem2 = !(getenv("SSG_SMEM_KERNEL") && !strcmp(getenv("SSG_SMEM_KERNEL"), "lane"));
	int scap = max_len + 2;
	if (smem2) CHK(ssg_seed_smem2(idx, opt, n_reads, d_seq, d_off, max_len, cap, d_intv, d_n, n_extend, (unsigned int)env_int("SSG_SMEM_MAX_EXT", 13 * max_len + 50), (uint32_t*)0));   /* 2000 at 150 bases: 1.8 % of the bench's reads go to the wave-per-read kernel (k_smem2.h) */
	else {
		const long nthreads = std::min<long>(((long)n_reads + block - 1) / block * block, 256L * env_int("SSG_SMEM_WAVES_PER_CU", 16) * 64);
		dbuf<ssg_intv_t> scratch((size_t)nthreads * 3 * scap + 64);
		CHKA(scratch);
		SSG_LAUNCH(ssg_k_smem_lane, nthreads / block, block, 0, idx->v, *opt, n_reads, (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, scratch.p, scap, n_extend);
	}
	CHK(rt_sync());
	{ unsigned int cc[5]; CHK(dev_class_counts(d_n, n_reads, 0, 0, 0, cc));
	  if (!cc[3]) { if (smem2) CHK(launch_smem_sort(n_reads, d_intv, d_n, cap)); return 0; } }
	std::vector<int32_t> hn(n_reads);
	CHK(rt_d2h(hn.data(), d_n, (size_t)n_reads * 4));
	std::vector<int32_t> ovf;
	for (int r = 0; r < n_reads; ++r) if (hn[r] < 0) ovf.push_back(r);
	if (ovf.empty()) { if (smem2) CHK(launch_smem_sort(n_reads, d_intv, d_n, cap)); return 0; }
	/* slow path: worst case is O(len^2) intervals in theory; len*8 has never been observed to overflow */
	int bigcap = max_len * 8 + 64, no = (int)ovf.size();
	dbuf<int32_t> d_ids(no), d_n2(no); dbuf<ssg_intv_t> d_big((size_t)no * bigcap);
	CHKA(d_ids); CHKA(d_n2); CHKA(d_big);
	CHK(d_ids.up(ovf.data(), no));
	long nt2 = ((long)no + block - 1) / block * block;
	dbuf<ssg_intv_t> scratch2((size_t)nt2 * 3 * bigcap);
	CHKA(scratch2);
	SSG_LAUNCH(ssg_k_smem_lane, nt2 / block, block, 0, idx->v, *opt, no, d_ids.p, d_seq, d_off, d_big.p, d_n2.p, bigcap, scratch2.p, bigcap, n_extend);
	CHK(rt_sync());
	std::vector<int32_t> hn2(no);
	CHK(d_n2.down(hn2.data(), no));
	for (int i = 0; i < no; ++i) if (hn2[i] < 0) { ssg_err_msg = "SMEM interval list exceeds 8 x read length"; return SSG_EOVERFLOW; }
	for (int i = 0; i < no; ++i) if (hn2[i] > cap) {   /* the dense per-read layout is too narrow for this batch: the caller widens it and calls again */
		if (!need_cap) { ssg_err_msg = "SMEM interval list exceeds the per-read capacity"; return SSG_EOVERFLOW; }
		*need_cap = std::max(*need_cap, (int)hn2[i]);
	}
	if (need_cap && *need_cap > cap) return 0;
	for (int i = 0; i < no; ++i) {
		std::vector<ssg_intv_t> tmp(hn2[i]);
		CHK(rt_d2h(tmp.data(), d_big.p + (size_t)i * bigcap, (size_t)hn2[i] * sizeof(ssg_intv_t)));
		CHK(rt_h2d(d_intv + (size_t)ovf[i] * cap, tmp.data(), (size_t)hn2[i] * sizeof(ssg_intv_t)));
		CHK(rt_h2d(d_n + ovf[i], &hn2[i], 4));
	}
	if (smem2) CHK(launch_smem_sort(n_reads, d_intv, d_n, cap));
	return 0;
}

int ssg_smem_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *seq, const int64_t *off,
                   int cap, ssg_intv_t *out_intv, int32_t *out_n)
{
	CHK(need_device());
	if (n_reads <= 0) return 0;
	int max_len = 0; for (int r = 0; r < n_reads; ++r) max_len = std::max<int>(max_len, (int)(off[r+1] - off[r]));
	dbuf<uint8_t> d_seq((size_t)off[n_reads] + 1); dbuf<int64_t> d_off(n_reads + 1); dbuf<ssg_intv_t> d_intv((size_t)n_reads * cap); dbuf<int32_t> d_n(n_reads);
	CHKA(d_seq); CHKA(d_off); CHKA(d_intv); CHKA(d_n);
	CHK(d_seq.up(seq, off[n_reads])); CHK(d_off.up(off, n_reads + 1));
	CHK(run_smem(idx, opt, n_reads, d_seq.p, d_off.p, max_len, cap, d_intv.p, d_n.p));
	CHK(d_intv.down(out_intv, (size_t)n_reads * cap));
	return d_n.down(out_n, n_reads);
}

/* permutation of 0..n-1 by descending key (counting sort on the clipped key; stable) */
static void order_desc(const std::vector<int32_t> &key, std::vector<int32_t> &order)
{
	const int K = 1 << 16;
	std::vector<int64_t> cnt(K + 1, 0);
	for (int32_t k : key) ++cnt[K - 1 - std::min(std::max(k, 0), K - 1)];
	int64_t s = 0;
	for (int i = 0; i < K; ++i) { int64_t c = cnt[i]; cnt[i] = s; s += c; }
	order.resize(key.size());
	for (size_t i = 0; i < key.size(); ++i) order[cnt[K - 1 - std::min(std::max(key[i], 0), K - 1)]++] = (int32_t)i;
}

/* radix sort of 64-bit keys on bits [begin_bit, end_bit) (stable) */
static int sort_keys_u64(uint64_t *k_in, uint64_t *k_out, long n, int begin_bit, int end_bit)
{
	if (n <= 0) return 0;
#ifdef SSG_EMU
	const uint64_t mask = (end_bit >= 64 ? ~0ull : (1ull << end_bit) - 1) & ~((1ull << begin_bit) - 1);
	std::vector<uint64_t> v(k_in, k_in + n);
	std::stable_sort(v.begin(), v.end(), [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
	memcpy(k_out, v.data(), (size_t)n * 8);
	return 0;
#else
	size_t tmp_bytes = 0;
	if (rocprim::radix_sort_keys(nullptr, tmp_bytes, k_in, k_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit) != hipSuccess) { ssg_err_msg = "rocprim radix_sort_keys (size query) failed"; return SSG_EHIP; }
	dbuf<uint8_t> tmp(tmp_bytes);
	CHKA(tmp);
	if (rocprim::radix_sort_keys(tmp.p, tmp_bytes, k_in, k_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, ssg_stream) != hipSuccess) { ssg_err_msg = "rocprim radix_sort_keys failed"; return SSG_EHIP; }
	return 0;
#endif
}

/* exclusive prefix sum of n int32 counts into n+1 int64 offsets, on the device; *total = out[n] */
struct ssg_to_i64 { SSG_DEVMEM int64_t operator()(int32_t v) const { return (int64_t)v; } };
static int dev_exclusive_scan(const int32_t *d_in, int64_t *d_out, long n, int64_t *total)
{
	if (n <= 0) { int64_t z = 0; CHK(rt_h2d(d_out, &z, 8)); *total = 0; return 0; }
#ifdef SSG_EMU
	int64_t t = 0; for (long i = 0; i < n; ++i) { d_out[i] = t; t += d_in[i]; } d_out[n] = t; *total = t;
	return 0;
#else
	rocprim::transform_iterator<const int32_t*, ssg_to_i64, int64_t> it(d_in, ssg_to_i64());
	size_t tmp_bytes = 0;
	if (rocprim::exclusive_scan(nullptr, tmp_bytes, it, d_out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>()) != hipSuccess) { ssg_err_msg = "rocprim exclusive_scan (size query) failed"; return SSG_EHIP; }
	dbuf<uint8_t> tmp(tmp_bytes + 16);
	CHKA(tmp);
	if (rocprim::exclusive_scan(tmp.p, tmp_bytes, it, d_out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), ssg_stream) != hipSuccess) { ssg_err_msg = "rocprim exclusive_scan failed"; return SSG_EHIP; }
	SSG_LAUNCH(ssg_k_scan_tail, 1, 64, 0, d_in, d_out, n);
	return rt_d2h(total, d_out + n, 8);
#endif
}

/* out[i] = max(in[0..i]) */
#ifndef SSG_EMU
struct ssg_max_i32 { __device__ int32_t operator()(const int32_t &a, const int32_t &b) const { return a > b ? a : b; } };
#endif
static int dev_scan_max_i32(const int32_t *d_in, int32_t *d_out, long n)
{
	if (n <= 0) return 0;
#ifdef SSG_EMU
	int32_t m = d_in[0]; for (long i = 0; i < n; ++i) { m = d_in[i] > m ? d_in[i] : m; d_out[i] = m; }
	return 0;
#else
	size_t tmp_bytes = 0;
	if (rocprim::inclusive_scan(nullptr, tmp_bytes, d_in, d_out, (size_t)n, ssg_max_i32()) != hipSuccess) { ssg_err_msg = "rocprim inclusive_scan (size query) failed"; return SSG_EHIP; }
	dbuf<uint8_t> tmp(tmp_bytes + 16);
	CHKA(tmp);
	if (rocprim::inclusive_scan(tmp.p, tmp_bytes, d_in, d_out, (size_t)n, ssg_max_i32(), ssg_stream) != hipSuccess) { ssg_err_msg = "rocprim inclusive_scan failed"; return SSG_EHIP; }
	return 0;
#endif
}

/* order[] = indices 0..n-1 by descending key (heaviest-first work lists), on the device */
static int dev_order_desc(const int32_t *d_key, int32_t *d_order, long n)
{
	if (n <= 0) return 0;
#ifdef SSG_EMU
	std::vector<int32_t> key(d_key, d_key + n), ord;
	order_desc(key, ord);
	memcpy(d_order, ord.data(), (size_t)n * 4);
	return 0;
#else
	dbuf<int32_t> iota(n), kout(n);
	CHKA(iota); CHKA(kout);
	SSG_LAUNCH(ssg_k_iota, (n + 255) / 256, 256, 0, iota.p, n);
	size_t tmp_bytes = 0;
	if (rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, d_key, kout.p, iota.p, d_order, (size_t)n, 0u, 32u) != hipSuccess) { ssg_err_msg = "rocprim radix_sort_pairs_desc (size query) failed"; return SSG_EHIP; }
	dbuf<uint8_t> tmp(tmp_bytes + 16);
	CHKA(tmp);
	if (rocprim::radix_sort_pairs_desc(tmp.p, tmp_bytes, d_key, kout.p, iota.p, d_order, (size_t)n, 0u, 32u, ssg_stream) != hipSuccess) { ssg_err_msg = "rocprim radix_sort_pairs_desc failed"; return SSG_EHIP; }
	return rt_sync();   /* the temporaries are released on return */
#endif
}

/* class counts of a device int32 array (ssg_k_class_counts) brought to the host */
static int dev_class_counts(const int32_t *d_key, long n, int tA, int tB, int tC, unsigned int out[5])
{
	dbuf<unsigned int> d_c(8);
	CHKA(d_c); CHK(d_c.zero());
	if (n > 0) SSG_LAUNCH(ssg_k_class_counts, (n + 255) / 256, 256, 0, d_key, n, tA, tB, tC, d_c.p);
	return d_c.down(out, 5);
}

/* ------------------------------- mem_align1_core for a batch ------------------------------- */
struct align1_dev_t {	/* device-resident result of stages 1-4 */
	dbuf<int64_t> seed_off; dbuf<ssg_alnreg_t> regs; dbuf<int32_t> n_reg;
	dbuf<uint8_t> sdp_fixed;   /* per read: its region list is a fixed point of the redundancy scan (no patch alignment was involved) */
	std::vector<int64_t> h_seed_off; int64_t tot_seeds;
};

static int run_align1(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *d_seq, const int64_t *d_off, int max_len,
                      align1_dev_t &o, uint64_t stats[8])
{
	/* intervals per read in the dense layout: upstream's list is unbounded; a batch that needs more widens the layout for itself
	 * and for the calls after it (low-complexity reads collect > len / 2 intervals from the re-seeding passes).  The layout starts at 5/4 of the
	 * read length: at len / 2 (until round 6) one batch in three or four of the 1 M-pair bench met a read beyond it and paid the one-lane kernel of
	 * the slow path (24-36 ms with the chip idle) plus a second run of the whole seeding stage (44-52 ms) -- profiles/r05_kernel_stats.csv shows the
	 * fourth ssg_k_smem2 launch in three steps; only the entries a read has are ever touched, so the width costs HBM (12 GB per million pairs), not time */
	static std::atomic<int> learned_cap(0);
	int cap = std::max(std::max(96, (max_len * 5 / 4 + 31) / 32 * 32), learned_cap.load());
	{ const int e = env_int("SSG_SMEM_CAP", 0); if (e > 0) cap = e; }   /* tests: a narrow layout walks the widening path */
	dbuf<ssg_intv_t> d_intv; dbuf<int32_t> d_nintv(n_reads), d_nseed(n_reads);
	CHKA(d_nintv); CHKA(d_nseed);
	dbuf<unsigned long long> d_next(1);
	CHKA(d_next);
	/* One call at a time in the seeding stage of a device.  Two calls in flight (bin/bwa: a lane each) gain only where their stages differ -- seeding is bound by
	 * the memory, extension by the vector units, the wave kernels by latency -- and lose nothing but the overlap when they run the same stage side by side; runs
	 * of the 8 M-pair script leg fell into two groups, 2.7 s and 3.7 s of alignment, by whether the two lanes happened to stay out of step.  A call that finds the
	 * other one seeding waits here once, and from then on they alternate.  Off unless SSG_SEED_TOKEN=1: on a second box every run was of the fast kind with and without it (profiles/r06y_literal_seed_token.json). */
	static std::mutex seed_token[16];
	std::unique_lock<std::mutex> seed_lock(seed_token[ssg_cur_dev & 15], std::defer_lock);
	if (env_int("SSG_SEED_TOKEN", 0) != 0) seed_lock.lock();
	for (;;) {
		int need = 0;
		if (!d_intv.alloc((size_t)n_reads * cap)) { ssg_err_msg = "device allocation failed: d_intv"; return SSG_ENOMEM; }
		CHK(d_next.zero());
		CHK(run_smem(idx, opt, n_reads, d_seq, d_off, max_len, cap, d_intv.p, d_nintv.p, d_next.p, &need));
		if (need <= cap) break;
		cap = (need + 31) / 32 * 32;
		{ int seen = learned_cap.load(); while (seen < cap && !learned_cap.compare_exchange_weak(seen, cap)) {} }
		if (ssg_debug()) fprintf(stderr, "[ssgpu] SMEM interval capacity widened to %d per read\n", cap);
	}
	if (stats) { unsigned long long c; CHK(d_next.down(&c, 1)); stats[5] = c; }
	if (seed_lock.owns_lock()) { CHK(rt_sync()); seed_lock.unlock(); }
	STAGE("smem");
	const int block = 256;
	dbuf<int32_t> d_pre;   /* per interval: occurrences of the read's earlier ones (ssg_k_sal finds a seed's interval by bisection); SSG_SAL_PREFIX=0: the walk along the list */
	if (env_int("SSG_SAL_PREFIX", 1) && !d_pre.alloc((size_t)n_reads * cap)) { ssg_err_msg = "device allocation failed: d_pre"; return SSG_ENOMEM; }
	SSG_LAUNCH(ssg_k_sal_count, (n_reads + block - 1) / block, block, 0, *opt, n_reads, d_intv.p, d_nintv.p, cap, d_nseed.p, d_pre.p);
	CHK(rt_sync());
	STAGE("sal_count");
	int64_t tot = 0;
	if (!o.seed_off.alloc(n_reads + 1)) { ssg_err_msg = "device allocation failed: seed_off"; return SSG_ENOMEM; }
	CHK(dev_exclusive_scan(d_nseed.p, o.seed_off.p, n_reads, &tot));
	o.tot_seeds = tot; o.h_seed_off.clear();
	size_t ts = (size_t)tot + 1;
	dbuf<ssg_seed_t> d_seeds(ts); dbuf<int32_t> d_srid(ts), d_order(ts), d_kept(ts), d_cseeds(ts), d_nchain(n_reads), d_err(n_reads);
	dbuf<ssg_chain_t> d_chains(ts); dbuf<uint64_t> d_srt(ts); dbuf<unsigned long long> d_cells(1);
	CHKA(d_seeds); CHKA(d_srid); CHKA(d_order); CHKA(d_kept); CHKA(d_cseeds); CHKA(d_nchain); CHKA(d_err); CHKA(d_chains); CHKA(d_srt); CHKA(d_cells);
	if (!o.regs.alloc(ts) || !o.n_reg.alloc(n_reads)) { ssg_err_msg = "device allocation failed: regs"; return SSG_ENOMEM; }
	CHK(d_cells.zero());
	{
		long g = (long)tot;
		/* every seed's read by a running maximum over marks at the reads' first seeds (d_order / d_kept are free until the chaining stage); SSG_SAL_READ_OF=0: bisection of seed_off by every lane */
		const bool rof = g > 0 && env_int("SSG_SAL_READ_OF", 1) != 0;
		if (rof) {
			CHK(rt_memset(d_order.p, 0, (size_t)g * 4));
			SSG_LAUNCH(ssg_k_sal_mark, (n_reads + block - 1) / block, block, 0, n_reads, o.seed_off.p, d_order.p);
			CHK(dev_scan_max_i32(d_order.p, d_kept.p, g));
		}
		SSG_LAUNCH(ssg_k_sal, (g + block - 1) / block, block, 0, idx->v, *opt, n_reads, d_intv.p, d_nintv.p, cap, o.seed_off.p, d_seeds.p, d_srid.p, (const int32_t*)d_pre.p, rof ? (const int32_t*)d_kept.p : (const int32_t*)0);
		{ dbuf<int32_t> done; done.swap(d_pre); }   /* 1.5 GB per million pairs back to the lane's arena (whatever takes it next is queued behind the kernel above) */
	}
	STAGE("sal");
	/* heaviest-first work order (seed count): the per-read cost of chaining / extension is heavy-tailed */
	dbuf<int32_t> d_work(n_reads); dbuf<unsigned int> d_queue(8);
	CHKA(d_work); CHKA(d_queue);
	CHK(dev_order_desc(d_nseed.p, d_work.p, n_reads)); CHK(d_queue.zero());
	if (ssg_debug() >= 2) { /* tuning: seeds-per-read histogram (power-of-two bins) */
		std::vector<int32_t> hns(n_reads);
		CHK(d_nseed.down(hns.data(), n_reads));
		long cnt[20] = {0}, sum[20] = {0};
		for (int r = 0; r < n_reads; ++r) { int b = 0; while ((1 << b) <= hns[r] && b < 19) ++b; ++cnt[b]; sum[b] += hns[r]; }
		for (int b = 0; b < 20; ++b) if (cnt[b]) fprintf(stderr, "[ssg] seeds/read < %d: %ld reads, %ld seeds\n", 1 << b, cnt[b], sum[b]);
	}
	{	/* d_work is heaviest first.  Reads with >= T seeds chain one wave each with their state in LDS, in four size classes that run
		 * concurrently (up to 5120, 2048, 1024, 512, 256 seeds = chains: 155 / 62 / 31 / 16 / 8 KB); the light rest one lane per read; the (very
		 * few) reads beyond 5120 seeds one lane each on their own stream.  The LDS kernels keep contig ids in 16 bits: an index with
		 * more contigs chains every read in the lane kernel.  SSG_CHAIN_WAVE_BIG = n sends every wave-class read with more than n seeds
		 * to the top class (tests); SSG_CHAIN_RANKED = 0 selects the array-shifting form of the insertion (A/B, tests). */
		const int T = idx->v.n_ctg > 32767 ? 1 << 30 : std::max(1, env_int("SSG_CHAIN_WAVE_MIN", 48));
		const int TB = env_int("SSG_CHAIN_WAVE_BIG", 0) > 0 ? env_int("SSG_CHAIN_WAVE_BIG", 0) : 1 << 30;
		const bool ranked = env_int("SSG_CHAIN_RANKED", 1) != 0;
		const int wsort = (env_int("SSG_CHAIN_WSORT", 1) ? 1 : 0) | (env_int("SSG_CHAIN_SPEC", 1) ? 2 : 0) | (env_int("SSG_CHAIN_BFLT", 1) ? 4 : 0);   /* the wave kernels' weight sort by the whole wave (k_chainw.h wv_introsort_whi) insertion 64 seeds a round and filter 64 chains a round; 0: by one lane / seed by seed / chain by chain (A/B, tests) */
		const int cap_lim = env_int("SSG_CHAIN_CAP_TEST", 1 << 30);   /* tests: pretend the ranked form holds fewer chains, to walk its fall-back (the shifting form) */
		int g[7], gl[3];   /* gl: reads with more than 63 / 31 / 15 seeds (the classes of the light reads' LDS kernel) */
		{	/* "greater than" counts of the seeds-per-read array (d_work sorts it descending: a binary search per threshold; thresholds descending: the counts ascend) */
			ssg_thr6_t th = { { 1 << 30, 5120, std::min(2048, TB), std::min(1024, TB), std::min(512, TB), std::min(256, TB), T - 1, 63, 31, 15 } };
			dbuf<unsigned int> d_c(16); unsigned int c[10];
			CHKA(d_c); CHK(d_c.zero());
			SSG_LAUNCH(ssg_k_count_gt6, 1, 64, 0, d_nseed.p, d_work.p, (long)n_reads, th, d_c.p);
			CHK(d_c.down(c, 10));
			for (int i = 0; i < 7; ++i) g[i] = (int)std::min(c[i], c[6]);
			for (int i = 1; i < 7; ++i) g[i] = std::max(g[i], g[i - 1]);
			for (int i = 0; i < 3; ++i) gl[i] = (int)c[7 + i];
		}
		const int n_heavy = g[6];
		const int nC = g[1], n5120 = g[2] - g[1], n2048 = g[3] - g[2], n1024 = g[4] - g[3], n512 = g[5] - g[4], n256 = g[6] - g[5];
		const int dbgp = ssg_debug() >= 2 ? -1 : 0;
		/* reads whose chains may lie differently in upstream's B-tree than in the kernels' (pos, sec) order -- more than 9 chains AND two at one position -- are flagged by
		 * every chaining kernel and chained again on the tree itself afterwards (k_chain.h ssg_k_chain_kb); SSG_CHAIN_KBTREE=0 leaves them as they are (the array order: tests) */
		dbuf<int32_t> d_kbflag((size_t)n_reads);
		CHKA(d_kbflag); CHK(d_kbflag.zero());
		int32_t *const kbf = env_int("SSG_CHAIN_KBTREE", 1) ? d_kbflag.p : (int32_t*)0;
		/* an error return between the fork and the join below hands this scope's buffers back to the arena: not before the side streams' kernels, which write them, are done */
		struct fork_guard_t { int n; bool armed; ~fork_guard_t() { if (armed) { ssg_join(n); (void)rt_sync(); } } } fork_guard = { 3, true };
		ssg_fork(3);
		{	/* The light reads go first, on a stream of their own, while this one ranks the heavy reads' seeds (small latency-bound launches that leave the chip idle); behind the wave
			 * kernels they would wait for LDS (the 63-seed class asks for 94 KB a workgroup) and run as a tail.  Heaviest first: up to 15 / 31 / 63 seeds with the read's state in the lane's part of LDS (k_chain.h), the rest -- and everything when the
			 * index has too many contigs for 14 bits, a read is too long for 9-bit query coordinates, or SSG_CHAIN_LDS = 0 (A/B, tests) -- on global memory */
			const bool use_lds = env_int("SSG_CHAIN_LDS", 1) != 0 && idx->v.n_ctg < 0x3fff && max_len < 512;
			const int r0 = n_heavy;
			const int b63 = use_lds ? std::min(n_reads, std::max(r0, gl[0])) : n_reads, b31 = std::min(n_reads, std::max(b63, gl[1])), b15 = std::min(n_reads, std::max(b31, gl[2]));
			if (b63 > r0) SSG_LAUNCH_ON(2, ssg_k_chain, (b63 - r0 + 63) / 64, 64, 0, idx->v, *opt, r0, b63, d_off, d_intv.p, d_nintv.p, cap, o.seed_off.p, d_seeds.p, d_srid.p,
			                   d_chains.p, d_order.p, d_kept.p, d_cseeds.p, d_nchain.p, dbgp, d_work.p, kbf);
#define SSG_CL_LAUNCH(CC, LN, from, to) do { if ((to) > (from)) SSG_LAUNCH_ON(2, (ssg_k_chain_lds<CC, LN>), ((to) - (from) + (LN) - 1) / (LN), (LN), 0, idx->v, *opt, (from), (to), d_off, d_intv.p, d_nintv.p, cap, o.seed_off.p, d_seeds.p, d_srid.p, \
			d_chains.p, d_order.p, d_cseeds.p, d_nchain.p, d_work.p, kbf); } while (0)
			if (env_int("SSG_CHAIN_LDS64_LANES", 32) == 32) SSG_CL_LAUNCH(64, 32, b63, b31); else SSG_CL_LAUNCH(64, 64, b63, b31);   /* 47 KB a workgroup instead of 94: fits beside the wave kernels' blocks */
			SSG_CL_LAUNCH(32, 64, b31, b15);
			SSG_CL_LAUNCH(16, 64, b15, n_reads);
#undef SSG_CL_LAUNCH
		}
		/* position ranks of the heavy reads' seeds: one stable radix sort of (read, reference position) over all of them */
		dbuf<uint16_t> d_hrank; dbuf<int64_t> d_hoff;
		if (ranked && n_heavy > nC) {
			dbuf<int32_t> d_hns(n_heavy);
			CHKA(d_hns);
			if (!d_hoff.alloc((size_t)n_heavy + 1)) { ssg_err_msg = "device allocation failed: d_hoff"; return SSG_ENOMEM; }
			SSG_LAUNCH(ssg_k_chw_count, (n_heavy + 255) / 256, 256, 0, n_heavy, d_work.p, o.seed_off.p, d_hns.p);
			int64_t n_hs = 0;
			CHK(dev_exclusive_scan(d_hns.p, d_hoff.p, n_heavy, &n_hs));
			if (n_hs >= (1LL << 32)) { ssg_err_msg = "more than 2^32 seeds in repeat-heavy reads of one call"; return SSG_EOVERFLOW; }
			dbuf<uint64_t> d_hk((size_t)n_hs + 1), d_hks((size_t)n_hs + 1); dbuf<uint32_t> d_hv((size_t)n_hs + 1), d_hvs((size_t)n_hs + 1);
			CHKA(d_hk); CHKA(d_hks); CHKA(d_hv); CHKA(d_hvs);
			if (!d_hrank.alloc((size_t)n_hs + 1)) { ssg_err_msg = "device allocation failed: d_hrank"; return SSG_ENOMEM; }
			SSG_LAUNCH(ssg_k_chw_keys, n_heavy, 64, 0, n_heavy, d_work.p, o.seed_off.p, d_seeds.p, d_hoff.p, d_hk.p, d_hv.p);
			CHK(sort_pairs_u64(d_hk.p, d_hks.p, d_hv.p, d_hvs.p, (long)n_hs));
			SSG_LAUNCH(ssg_k_chw_ranks, (n_hs + 255) / 256, 256, 0, (long)n_hs, d_hks.p, d_hvs.p, d_hoff.p, d_hrank.p);
			CHK(rt_sync());
		}
		const uint16_t *hr = ranked ? d_hrank.p : (const uint16_t*)0; const int64_t *ho = ranked ? d_hoff.p : (const int64_t*)0;
		/* the classes are independent and each of the heavy ones fills a fraction of the chip: overlap them */
		/* four queues: the two big classes one each, the short jobs (top class, small classes, then the one-lane monsters) in a row on the third,
		 * the light reads' lane kernel on the default stream (more streams than hardware queues serialize behind one another anyway) */
		ssg_fork(2);   /* (again, the wave kernels' two streams: they read the ranks) */
		int r0 = nC;
#define SSG_CHW_LAUNCH(si, CC, cnt, maxwg, qi) do { if ((cnt) > 0) SSG_LAUNCH_ON(si, ssg_k_chain_wave<CC>, std::min((int)(cnt), (int)(maxwg)), 64, 0, idx->v, *opt, r0, r0 + (cnt), d_off, d_intv.p, d_nintv.p, cap, \
		o.seed_off.p, d_seeds.p, d_srid.p, d_chains.p, d_order.p, d_cseeds.p, d_nchain.p, d_work.p, d_queue.p + (qi), kbf, hr, ho, std::min((int)(CC), cap_lim), wsort); r0 += (cnt); } while (0)
		SSG_CHW_LAUNCH(0, 5120, n5120, 256, 1);
		SSG_CHW_LAUNCH(1, 2048, n2048, 512, 3);
		SSG_CHW_LAUNCH(1, 1024, n1024, 1280, 2);   /* (behind the 2048 class: this stream + two side streams + the light reads' stream are the four hardware queues; a fifth stream shares one) */
		SSG_CHW_LAUNCH(0, 512, n512, 2560, 5);
		SSG_CHW_LAUNCH(0, 256, n256, 5120, 4);
#undef SSG_CHW_LAUNCH
		if (nC) SSG_LAUNCH_ON(0, ssg_k_chain, (nC + 63) / 64, 64, 0, idx->v, *opt, 0, nC, d_off, d_intv.p, d_nintv.p, cap, o.seed_off.p, d_seeds.p, d_srid.p,
		                   d_chains.p, d_order.p, d_kept.p, d_cseeds.p, d_nchain.p, dbgp, d_work.p, kbf);
		ssg_join(3);
		fork_guard.armed = false;
		if (kbf) {   /* the flagged reads (none in a million simulated human pairs; the constructed reads of tests/test_chain_btree.py) on klib's B-tree */
			dbuf<int32_t> d_klist((size_t)n_reads), d_kneed((size_t)n_reads); dbuf<unsigned int> d_nk(1);
			CHKA(d_klist); CHKA(d_kneed); CHKA(d_nk); CHK(d_nk.zero());
			SSG_LAUNCH(ssg_k_chain_kb_list, (n_reads + 255) / 256, 256, 0, n_reads, (const int32_t*)d_kbflag.p, o.seed_off.p, d_klist.p, d_kneed.p, d_nk.p);
			unsigned int nk = 0;
			CHK(d_nk.down(&nk, 1));
			if (stats) stats[7] = nk;
			if (nk) {
				std::vector<int32_t> kl(nk), kn(nk);   /* (the list in read order: the kernel's atomics hand out places as they come) */
				CHK(d_klist.down(kl.data(), nk)); CHK(d_kneed.down(kn.data(), nk));
				std::vector<size_t> by(nk); for (size_t i = 0; i < nk; ++i) by[i] = i;
				std::sort(by.begin(), by.end(), [&](size_t a, size_t b) { return kl[a] < kl[b]; });
				std::vector<int32_t> kl2(nk); std::vector<int64_t> ko((size_t)nk + 1, 0);
				for (size_t i = 0; i < nk; ++i) { kl2[i] = kl[by[i]]; ko[i + 1] = ko[i] + kn[by[i]]; }
				dbuf<int32_t> d_slab((size_t)ko[nk] + 1), d_kerr(1); dbuf<int64_t> d_ko((size_t)nk + 1);
				CHKA(d_slab); CHKA(d_kerr); CHKA(d_ko); CHK(d_kerr.zero());
				CHK(d_klist.up(kl2.data(), nk)); CHK(d_ko.up(ko.data(), (size_t)nk + 1));
				SSG_LAUNCH(ssg_k_chain_kb, (nk + 63) / 64, 64, 0, idx->v, *opt, (int)nk, (const int32_t*)d_klist.p, d_off, d_intv.p, d_nintv.p, cap, o.seed_off.p, d_seeds.p, d_srid.p,
				           d_chains.p, d_order.p, d_kept.p, d_cseeds.p, d_nchain.p, d_slab.p, (const int64_t*)d_ko.p, d_kerr.p);
				int32_t ke = 0;
				CHK(d_kerr.down(&ke, 1));
				if (ke) { ssg_err_msg = "chaining on the B-tree exceeded its node slab"; return SSG_EOVERFLOW; }
				if (ssg_debug()) fprintf(stderr, "[ssgpu] %u reads chained again on klib's B-tree (more than 9 chains and two at one position)\n", nk);
			}
		}
	}
	STAGE("chain");
	/* ---- extensions of every chain's first seed, one lane each (k_extlane.h) ---- */
	dbuf<int64_t> d_choff((size_t)n_reads + 1);
	CHKA(d_choff);
	int64_t n_jobs64 = 0;
	CHK(dev_exclusive_scan(d_nchain.p, d_choff.p, n_reads, &n_jobs64));
	if (n_jobs64 >= (int64_t)1 << 31) { ssg_err_msg = "more than 2^31 chains in one call"; return SSG_EOVERFLOW; }
	const long n_jobs = (long)n_jobs64;
	dbuf<ssg_xjob_t> d_xjobs((size_t)n_jobs + 1); dbuf<ssg_xres_t> d_xl((size_t)n_jobs + 1), d_xr((size_t)n_jobs + 1);
	dbuf<uint64_t> d_kl((size_t)n_jobs + 1), d_kr((size_t)n_jobs + 1), d_sl((size_t)n_jobs + 1), d_sr((size_t)n_jobs + 1);
	CHKA(d_xjobs); CHKA(d_xl); CHKA(d_xr); CHKA(d_kl); CHKA(d_kr); CHKA(d_sl); CHKA(d_sr);
	if (n_jobs > 0) {
		if (opt->a * 2 * max_len + 64 >= 8191) { ssg_err_msg = "match score x read length beyond the 13-bit DP cells of the extension kernel"; return SSG_EINVAL; }
		if (opt->min_chain_weight > 0 && 2.8f * (float)opt->min_chain_weight <= 0.05f * (float)max_len) {   /* upstream mem_flt_chained_seeds would run (MEM_HSP_COEF x W <= MEM_SEEDSW_COEF x l): its seed-level SW is not built */
			ssg_err_msg = "-W this small against this read length turns on upstream's chained-seed filter (long-read path), which this build does not have"; return SSG_EINVAL; }
		if (opt->a > 31 || opt->a < 0 || opt->b > 32 || opt->b < 0) { ssg_err_msg = "match score above 31 or mismatch penalty above 32: beyond the 6-bit score table of the extension kernel"; return SSG_EINVAL; }
		const int short_cap = 72;   /* sides up to 72 bases run with half the LDS per wave (two waves per SIMD) */
		dbuf<unsigned int> d_nlong(2);
		CHKA(d_nlong); CHK(d_nlong.zero());
		SSG_LAUNCH(ssg_k_ext_prep, (n_jobs + 255) / 256, 256, 0, idx->v, *opt, n_reads, n_jobs, d_off, o.seed_off.p, d_seeds.p, d_chains.p, d_order.p, d_cseeds.p,
		           d_choff.p, (int)SSG_TWIN_GLB, d_xjobs.p, d_kl.p, d_kr.p, short_cap, d_nlong.p, d_seq, env_int("SSG_EXT_ROWS_KEY", 1));
		CHK(sort_keys_u64(d_kl.p, d_sl.p, n_jobs, 32, 50)); CHK(sort_keys_u64(d_kr.p, d_sr.p, n_jobs, 32, 50));   /* 9 + 9 bits: 511 - side length, 511 - expected rows */
		unsigned int h_nlong[2];
		CHK(d_nlong.down(h_nlong, 2));
		if (ssg_debug()) fprintf(stderr, "[ssgpu] ext jobs %ld, long sides %u / %u\n", n_jobs, h_nlong[0], h_nlong[1]);
		/* Classes of 8 more columns each, a launch per class with the LDS its longest side needs (k_extlane.h ssg_k_ext_lane_dyn): the list is sorted longest side first, the
		 * class boundaries are binary searches; a side's classes go round three queues (no tail of one launch before the next starts, and a CU holds a mix of block sizes),
		 * the right sides after all left ones (they start from the left side's score).  SSG_EXT_DYN=0: the fixed classes below. */
		const bool ext_dyn = env_int("SSG_EXT_DYN", 1) != 0;
		if (ext_dyn) {
			ssg_thr64_t th; int caps[64], ncap = 0;
			caps[ncap++] = 40; caps[ncap++] = 72;
			for (int c = 80; c < max_len + 8 && c <= 320; c += 8) caps[ncap++] = c;
			th.n = ncap + 1;
			for (int k = 0; k < ncap; ++k) th.t[k] = 511 - caps[k];   /* #sides longer than caps[k] */
			th.t[ncap] = 511;                                        /* #sides longer than 0 */
			dbuf<unsigned int> d_b(128); unsigned int hb[2][64];
			CHKA(d_b);
			SSG_LAUNCH(ssg_k_sorted_hi_below, 1, 64, 0, d_sl.p, n_jobs, th, d_b.p);
			SSG_LAUNCH(ssg_k_sorted_hi_below, 1, 64, 0, d_sr.p, n_jobs, th, d_b.p + 64);
			CHK(d_b.down(&hb[0][0], 128));
#ifndef SSG_EMU
			if (max_len > 240) {   /* blocks above 64 KB */
				static bool attr_set[SSG_MAX_DEV] = { false };
				if (!attr_set[ssg_cur_dev]) { (void)hipFuncSetAttribute((const void*)ssg_k_ext_lane_dyn<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (320 + 8) * 256); attr_set[ssg_cur_dev] = true; }
			}
#endif
			for (int side = 0; side < 2; ++side) {
				const uint64_t *srt = side ? d_sr.p : d_sl.p;
				ssg_fork(2);
				for (int k = ncap - 1, q = 0; k >= 0; --k) {   /* longest class first */
					const long from = (long)hb[side][k], to = (long)hb[side][k == 0 ? ncap : k - 1];
					if (to <= from) continue;
					/* (eight columns per trip instead of four -- twice the LDS loads in flight for the classes that run one wave a SIMD -- was measured slower on the MI355X:
					 * 125.3 vs 119.4 ms at 2x150, 157.7 vs 151.3 at 2x250, profiles/r06f_ext_unroll_ab*.json: the cell loop waits for its own dependent VALU chain, not for LDS) */
					const int c = caps[k], U = c > 72 ? 4 : 2; const size_t lds = (size_t)(c + 2 * U) * 256;
#define SSG_XL_GO(LAUNCH, ...) do { if (U == 4) LAUNCH(__VA_ARGS__ ssg_k_ext_lane_dyn<4>, (to - from + 63) / 64, 64, lds, idx->v, *opt, side, from, to, srt, d_xjobs.p, d_seq, d_off, d_xl.p, d_xr.p, d_cells.p, c); \
                                    else LAUNCH(__VA_ARGS__ ssg_k_ext_lane_dyn<2>, (to - from + 63) / 64, 64, lds, idx->v, *opt, side, from, to, srt, d_xjobs.p, d_seq, d_off, d_xl.p, d_xr.p, d_cells.p, c); } while (0)
					if (q % 3 == 0) SSG_XL_GO(SSG_LAUNCH); else if (q % 3 == 1) SSG_XL_GO(SSG_LAUNCH_ON, 0,); else SSG_XL_GO(SSG_LAUNCH_ON, 1,);
#undef SSG_XL_GO
					++q;
				}
				ssg_join(2);
			}
		} else
		for (int side = 0; side < 2; ++side) {
			const uint64_t *srt = side ? d_sr.p : d_sl.p;
			const long nl = (long)h_nlong[side];   /* jobs are sorted longest side first */
			if (max_len <= 136 + opt->min_seed_len) {
				if (nl) SSG_LAUNCH(ssg_k_ext_lane<136>, (nl + 63) / 64, 64, 0, idx->v, *opt, side, 0L, nl, srt, d_xjobs.p, d_seq, d_off, d_xl.p, d_xr.p, d_cells.p);
				if (n_jobs > nl) SSG_LAUNCH(ssg_k_ext_lane<72>, (n_jobs - nl + 63) / 64, 64, 0, idx->v, *opt, side, nl, n_jobs, srt, d_xjobs.p, d_seq, d_off, d_xl.p, d_xr.p, d_cells.p);
			} else if (max_len <= 256) SSG_LAUNCH(ssg_k_ext_lane<256>, (n_jobs + 63) / 64, 64, 0, idx->v, *opt, side, 0L, n_jobs, srt, d_xjobs.p, d_seq, d_off, d_xl.p, d_xr.p, d_cells.p);
			else SSG_LAUNCH(ssg_k_ext_lane<320>, (n_jobs + 63) / 64, 64, 0, idx->v, *opt, side, 0L, n_jobs, srt, d_xjobs.p, d_seq, d_off, d_xl.p, d_xr.p, d_cells.p);
		}
	}
	STAGE("ext_lane");
	{
		const int wpb = SSG_WAVES_PER_WG;
		long nwg = std::min<long>(((long)n_reads + wpb - 1) / wpb, 256 * SSG_C2A_WAVES_PER_SIMD);
		dbuf<uint8_t> d_tglb((size_t)nwg * wpb * SSG_TWIN_GLB); dbuf<ssg_sdp_big_t> d_sdpbig((size_t)nwg * wpb); dbuf<ssg_alnreg_t> d_bcopy((size_t)nwg * wpb * SSG_SDP_BIG);
		CHKA(d_tglb); CHKA(d_sdpbig); CHKA(d_bcopy);
		/* light reads one lane each; what is left (long lists, later-seed extensions, patches) one wave each */
		dbuf<int32_t> d_todo((size_t)n_reads); dbuf<unsigned int> d_ntodo(1);
		CHKA(d_todo); CHKA(d_ntodo); CHK(d_ntodo.zero());
		if (!o.sdp_fixed.alloc((size_t)n_reads)) { ssg_err_msg = "device allocation failed: sdp_fixed"; return SSG_ENOMEM; }
		CHK(o.sdp_fixed.zero());
		SSG_LAUNCH(ssg_k_chain2aln_lane, (n_reads + 63) / 64, 64, 0, idx->v, *opt, n_reads, d_off, o.seed_off.p, d_seeds.p, d_cseeds.p, d_nchain.p, o.regs.p, o.n_reg.p, d_err.p,
		           d_choff.p, d_xjobs.p, d_xl.p, d_xr.p, d_work.p, d_todo.p, d_ntodo.p, o.sdp_fixed.p);
		SSG_LAUNCH_W(max_len > 255, ssg_k_chain2aln, nwg, wpb * 64, 0, idx->v, *opt, n_reads, d_seq, d_off, o.seed_off.p, d_seeds.p, d_chains.p, d_order.p, d_cseeds.p,
		           d_nchain.p, d_srt.p, o.regs.p, o.n_reg.p, d_tglb.p, d_err.p, d_cells.p, d_work.p, d_queue.p, ssg_debug() >= 2, d_sdpbig.p, d_bcopy.p,
		           d_choff.p, d_xjobs.p, d_xl.p, d_xr.p, d_todo.p, d_ntodo.p, o.sdp_fixed.p);
		CHK(rt_sync());
	}
	{ unsigned int cc[5]; CHK(dev_class_counts(d_err.p, n_reads, 0, 0, 0, cc)); if (cc[4]) { ssg_err_msg = "reference window of a chain exceeds SSG_TWIN_GLB"; return SSG_EOVERFLOW; } }
	if (stats) { unsigned long long c; CHK(d_cells.down(&c, 1)); stats[0] = (uint64_t)tot; stats[1] = c; stats[6] = (uint64_t)n_jobs; }
	return 0;
}

int ssg_seeds_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *seq, const int64_t *off,
                    int64_t *seed_off, ssg_seed_t **seeds, int32_t **rids)
{
	CHK(need_device());
	*seeds = 0; *rids = 0;
	if (n_reads <= 0) { if (seed_off) seed_off[0] = 0; return 0; }
	int max_len = 0; for (int r = 0; r < n_reads; ++r) max_len = std::max<int>(max_len, (int)(off[r+1] - off[r]));
	if (max_len > SSG_MAX_READ_LEN) { ssg_err_msg = "reads longer than " SSG_STR(SSG_MAX_READ_LEN) " bases are outside this build's scope"; return SSG_EINVAL; }
	dbuf<uint8_t> d_seq((size_t)off[n_reads] + 1); dbuf<int64_t> d_off(n_reads + 1), d_soff(n_reads + 1); dbuf<int32_t> d_nintv(n_reads), d_nseed(n_reads);
	CHKA(d_seq); CHKA(d_off); CHKA(d_soff); CHKA(d_nintv); CHKA(d_nseed);
	CHK(d_seq.up(seq, off[n_reads])); CHK(d_off.up(off, n_reads + 1));
	dbuf<ssg_intv_t> d_intv;
	for (int cap = std::max(64, max_len / 2); ; ) {   /* the dense interval layout widens itself, as in run_align1 */
		int need = 0;
		if (!d_intv.alloc((size_t)n_reads * cap)) { ssg_err_msg = "device allocation failed: d_intv"; return SSG_ENOMEM; }
		CHK(run_smem(idx, opt, n_reads, d_seq.p, d_off.p, max_len, cap, d_intv.p, d_nintv.p, 0, &need));
		if (need > cap) { cap = (need + 31) / 32 * 32; continue; }
		SSG_LAUNCH(ssg_k_sal_count, (n_reads + 255) / 256, 256, 0, *opt, n_reads, d_intv.p, d_nintv.p, cap, d_nseed.p, (int32_t*)0);
		int64_t tot = 0;
		CHK(dev_exclusive_scan(d_nseed.p, d_soff.p, n_reads, &tot));
		dbuf<ssg_seed_t> d_seeds((size_t)tot + 1); dbuf<int32_t> d_srid((size_t)tot + 1);
		CHKA(d_seeds); CHKA(d_srid);
		if (tot > 0) SSG_LAUNCH(ssg_k_sal, (tot + 255) / 256, 256, 0, idx->v, *opt, n_reads, d_intv.p, d_nintv.p, cap, d_soff.p, d_seeds.p, d_srid.p, (const int32_t*)0, (const int32_t*)0);
		CHK(rt_sync());
		CHK(d_soff.down(seed_off, (size_t)n_reads + 1));
		*seeds = (ssg_seed_t*)malloc(sizeof(ssg_seed_t) * (size_t)(tot + 1)); *rids = (int32_t*)malloc(4 * (size_t)(tot + 1));
		if (!*seeds || !*rids) { free(*seeds); free(*rids); *seeds = 0; *rids = 0; ssg_err_msg = "host allocation failed"; return SSG_ENOMEM; }
		CHK(d_seeds.down(*seeds, (size_t)tot)); CHK(d_srid.down(*rids, (size_t)tot));
		return 0;
	}
}

int ssg_align1_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *seq, const int64_t *off,
                     int64_t *reg_off, ssg_alnreg_t **regs, uint64_t stats[8])
{
	CHK(need_device());
	*regs = 0;
	if (n_reads <= 0) { if (reg_off) reg_off[0] = 0; return 0; }
	int max_len = 0; for (int r = 0; r < n_reads; ++r) max_len = std::max<int>(max_len, (int)(off[r+1] - off[r]));
	if (max_len > SSG_MAX_READ_LEN) { ssg_err_msg = "reads longer than " SSG_STR(SSG_MAX_READ_LEN) " bases are outside this build's scope"; return SSG_EINVAL; }
	dbuf<uint8_t> d_seq((size_t)off[n_reads] + 1); dbuf<int64_t> d_off(n_reads + 1);
	CHKA(d_seq); CHKA(d_off);
	CHK(d_seq.up(seq, off[n_reads])); CHK(d_off.up(off, n_reads + 1));
	align1_dev_t o;
	CHK(run_align1(idx, opt, n_reads, d_seq.p, d_off.p, max_len, o, stats));
	std::vector<int32_t> hn(n_reads);
	CHK(o.n_reg.down(hn.data(), n_reads));
	int64_t tot = 0;
	for (int r = 0; r < n_reads; ++r) { reg_off[r] = tot; tot += hn[r]; }
	reg_off[n_reads] = tot;
	ssg_alnreg_t *out = (ssg_alnreg_t*)malloc(sizeof(ssg_alnreg_t) * (size_t)(tot + 1));
	o.h_seed_off.resize((size_t)n_reads + 1);
	CHK(o.seed_off.down(o.h_seed_off.data(), (size_t)n_reads + 1));
	std::vector<ssg_alnreg_t> all((size_t)o.tot_seeds + 1);
	CHK(o.regs.down(all.data(), (size_t)o.tot_seeds));
	for (int r = 0; r < n_reads; ++r) memcpy(out + reg_off[r], all.data() + o.h_seed_off[r], sizeof(ssg_alnreg_t) * (size_t)hn[r]);
	*regs = out;
	return 0;
}

} /* extern "C" */

/* ======================= paired-end stage (rows a9-a12) ======================= */

/* upstream mem_pestat's statistics, from the per-orientation insert-size histogram.  The sorted
 * isize array upstream walks is the histogram read in bin order, so every double-precision sum is
 * accumulated in exactly upstream's order. */
static void host_pestat(const ssg_mem_opt_t *opt, const uint32_t *hist /* [4][SSG_MAX_INS_HIST] */, ssg_pestat_t pes[4])
{
	uint64_t n[4]; int d, max = 0;
	memset(pes, 0, 4 * sizeof(ssg_pestat_t));
	for (d = 0; d < 4; ++d) {
		const uint32_t *h = hist + (size_t)d * SSG_MAX_INS_HIST;
		ssg_pestat_t *r = &pes[d];
		n[d] = 0;
		for (int v = 0; v < SSG_MAX_INS_HIST; ++v) n[d] += h[v];
		if (n[d] < 10) { r->failed = 1; continue; }
		auto kth = [&](uint64_t k) { uint64_t c = 0; for (int v = 0; v < SSG_MAX_INS_HIST; ++v) { c += h[v]; if (c > k) return v; } return SSG_MAX_INS_HIST - 1; };
		int p25 = kth((uint64_t)(int)(.25 * n[d] + .499)), p75 = kth((uint64_t)(int)(.75 * n[d] + .499));
		r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
		uint64_t x = 0; r->avg = 0;
		for (int v = 0; v < SSG_MAX_INS_HIST; ++v) if (v >= r->low && v <= r->high) for (uint32_t c = 0; c < h[v]; ++c) { r->avg += v; ++x; }
		r->avg /= (int)x;
		r->std = 0;
		for (int v = 0; v < SSG_MAX_INS_HIST; ++v) if (v >= r->low && v <= r->high) for (uint32_t c = 0; c < h[v]; ++c) r->std += (v - r->avg) * (v - r->avg);
		r->std = sqrt(r->std / (int)x);
		r->low  = (int)(p25 - 3.0 * (p75 - p25) + .499);
		r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
		if (r->low  > r->avg - 4.0 * r->std) r->low  = (int)(r->avg - 4.0 * r->std + .499);
		if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
		if (r->low < 1) r->low = 1;
	}
	for (d = 0; d < 4; ++d) max = max > (int)n[d] ? max : (int)n[d];
	for (d = 0; d < 4; ++d) if (pes[d].failed == 0 && n[d] < max * 0.05) pes[d].failed = 1;
}

/* the whole PE hot path on device-resident inputs; `keep` != NULL leaves the records in HBM
 * instead of downloading them into `res` */
static thread_local unsigned int ssg_r2a_last[1 + SSG_R2D_CLASSES];   /* of this thread's last call: records left to the wave kernel, records per class of the lane DP */
/* CIGAR / NM / MD of the compacted requests (k_aln.h): gap-free records one lane each; records whose band is narrow one lane each through the DP (three classes of
 * band width, a launch each with the LDS that class needs, side by side); the rest -- wide bands, and records whose first alignment does not end upstream's
 * loop -- one wave each.  SSG_R2A_DPLANE=0: no lane DP (A/B, tests). */
static int run_reg2aln(const ssg_index *idx, const ssg_mem_opt_t *opt, int64_t nreq, const ssg_alnreq_t *d_creq, const ssg_alnreg_t *d_regs, const uint8_t *d_seq, const int64_t *d_off,
                       ssg_aln_t *d_alns, int32_t *d_gerr, unsigned long long *d_cnt, int max_len)
{
	const int wpb = SSG_WAVES_PER_WG;
	long nwg = std::min<long>(((long)nreq + wpb - 1) / wpb, SSG_MAX_RESIDENT_WG);
	long nw = nwg * wpb;
	dbuf<uint8_t> d_tglb((size_t)nw * SSG_TWIN_GLB), d_z((size_t)nw * SSG_Z_CAP);
	CHKA(d_tglb); CHKA(d_z);
	dbuf<int32_t> d_rtodo((size_t)nreq + 1); dbuf<unsigned int> d_nrtodo(1 + SSG_R2D_CLASSES);
	CHKA(d_rtodo); CHKA(d_nrtodo); CHK(d_nrtodo.zero());
	const bool dplane = env_int("SSG_R2A_DPLANE", 1) != 0 && max_len <= SSG_ALN_QLDS;
	dbuf<int32_t> d_dpl(dplane ? (size_t)nreq * SSG_R2D_CLASSES + 1 : 1);
	CHKA(d_dpl);
	SSG_LAUNCH(ssg_k_reg2aln_lane, (nreq + 63) / 64, 64, 0, idx->v, *opt, (long)nreq, d_creq, d_regs, d_seq, d_off, d_alns, d_gerr, d_rtodo.p, d_nrtodo.p,
	           dplane ? d_dpl.p : (int32_t*)0, d_nrtodo.p + 1);
	if (dplane) {
		const int qwords = (max_len + 7) / 8, zrows = max_len + 32;
		const long G = std::min<long>((nreq + 63) / 64, (long)env_int("SSG_R2D_WAVES", 2048));
		size_t zoff[SSG_R2D_CLASSES + 1]; zoff[0] = 0;
		for (int c = 0; c < SSG_R2D_CLASSES; ++c) { const int wm = c == 0 ? 8 : c == 1 ? 16 : 32; zoff[c + 1] = zoff[c] + (size_t)G * zrows * ((2 * wm + 1 + 7) / 8) * 64; }
		dbuf<uint32_t> d_zs(zoff[SSG_R2D_CLASSES] + 1);
		CHKA(d_zs);
		ssg_fork(2);
		for (int c = SSG_R2D_CLASSES - 1; c >= 0; --c) {   /* widest class first */
			const int wm = c == 0 ? 8 : c == 1 ? 16 : 32, zw = (2 * wm + 1 + 7) / 8;
			const size_t lds = (size_t)(2 * wm + 2 + qwords) * 256;
#define SSG_R2D_GO(LAUNCH, ...) LAUNCH(__VA_ARGS__ ssg_k_reg2aln_dplane, G, 64, lds, idx->v, *opt, d_dpl.p + (size_t)c * nreq, d_nrtodo.p + 1 + c, d_creq, d_regs, d_seq, d_off, d_alns, \
                                       d_zs.p + zoff[c], wm, qwords, zw, zrows, d_gerr, d_cnt, d_rtodo.p, d_nrtodo.p)
			if (c == 2) SSG_R2D_GO(SSG_LAUNCH); else if (c == 1) SSG_R2D_GO(SSG_LAUNCH_ON, 0,); else SSG_R2D_GO(SSG_LAUNCH_ON, 1,);
#undef SSG_R2D_GO
		}
		ssg_join(2);
		SSG_LAUNCH_W(max_len > 255, ssg_k_reg2aln, nwg, wpb * 64, 0, idx->v, *opt, (long)nreq, d_creq, d_regs, d_seq, d_off, d_alns, d_tglb.p, d_z.p, d_gerr, d_cnt, d_rtodo.p, d_nrtodo.p);
		CHK(rt_sync());   /* (before the slabs go back to the arena) */
		CHK(d_nrtodo.down(ssg_r2a_last, 1 + SSG_R2D_CLASSES));
		if (ssg_debug()) fprintf(stderr, "[ssgpu] reg2aln: %lld requests; lane DP by band class %u / %u / %u, wave kernel %u\n", (long long)nreq, ssg_r2a_last[1], ssg_r2a_last[2], ssg_r2a_last[3], ssg_r2a_last[0]);
		return 0;
	}
	SSG_LAUNCH_W(max_len > 255, ssg_k_reg2aln, nwg, wpb * 64, 0, idx->v, *opt, (long)nreq, d_creq, d_regs, d_seq, d_off, d_alns, d_tglb.p, d_z.p, d_gerr, d_cnt, d_rtodo.p, d_nrtodo.p);
	CHK(rt_sync());
	return 0;
}

extern "C" void ssg_dbg_reg2aln_counts(unsigned int out[4]) { for (int k = 0; k < 1 + SSG_R2D_CLASSES; ++k) out[k] = ssg_r2a_last[k]; }

static int pe_core(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *d_seq_p, const int64_t *d_off_p, int max_len,
                   const int32_t *d_pb_p, int n_batches, int64_t id0, const ssg_pestat_t *pes0, ssg_pe_result *res, pe_dev_t *keep)
{
	const int n_reads = 2 * n_pairs;
	struct { const uint8_t *p; } d_seq = { d_seq_p };
	struct { const int64_t *p; } d_off = { d_off_p };
	struct { const int32_t *p; } d_pb = { d_pb_p };
	res->n_reads = n_reads; res->n_batches = n_batches; memset(res->stats, 0, sizeof(res->stats));
	align1_dev_t a1;
	CHK(run_align1(idx, opt, n_reads, d_seq.p, d_off.p, max_len, a1, res->stats));
	const int block = 256;
	/* ---- insert-size statistics ---- */
	res->pes.resize((size_t)n_batches * 4);
	if (pes0) { for (int b = 0; b < n_batches; ++b) memcpy(&res->pes[(size_t)b * 4], pes0, 4 * sizeof(ssg_pestat_t)); }
	else {
		dbuf<uint32_t> d_hist((size_t)n_batches * 4 * SSG_MAX_INS_HIST);
		CHKA(d_hist); CHK(d_hist.zero());
		SSG_LAUNCH(ssg_k_pestat_hist, (n_pairs + block - 1) / block, block, 0, idx->v, *opt, n_pairs, a1.seed_off.p, a1.regs.p, a1.n_reg.p, d_pb.p, d_hist.p);
		CHK(rt_sync());
		std::vector<uint32_t> hh((size_t)n_batches * 4 * SSG_MAX_INS_HIST);
		CHK(d_hist.down(hh.data(), hh.size()));
		for (int b = 0; b < n_batches; ++b) host_pestat(opt, hh.data() + (size_t)b * 4 * SSG_MAX_INS_HIST, &res->pes[(size_t)b * 4]);
	}
	STAGE("pestat");
	dbuf<ssg_pestat_t> d_pes((size_t)n_batches * 4);
	CHKA(d_pes); CHK(d_pes.up(res->pes.data(), res->pes.size()));
	/* ---- pairing-stage region slices with head-room for rescued hits ---- */
	dbuf<int32_t> d_cap2(n_reads), d_capq(n_reads), d_pkey(n_pairs), d_pw(n_pairs);
	dbuf<int64_t> d_r2off(n_reads + 1), d_reqoff(n_reads + 1);
	CHKA(d_cap2); CHKA(d_capq); CHKA(d_pkey); CHKA(d_pw); CHKA(d_r2off); CHKA(d_reqoff);
	SSG_LAUNCH(ssg_k_pair_caps, (n_reads + block - 1) / block, block, 0, n_reads, a1.n_reg.p, opt->max_matesw, d_cap2.p, d_capq.p, d_pkey.p);
	int64_t t2 = 0, tq = 0;
	CHK(dev_exclusive_scan(d_cap2.p, d_r2off.p, n_reads, &t2)); CHK(dev_exclusive_scan(d_capq.p, d_reqoff.p, n_reads, &tq));
	CHK(dev_order_desc(d_pkey.p, d_pw.p, n_pairs));   /* heaviest-first pair order for the pairing-stage kernels (key: candidate regions of both ends) */
	dbuf<ssg_alnreg_t> d_regs2((size_t)t2 + 1); dbuf<int32_t> d_perr(n_pairs), d_zbuf((size_t)t2 + 1), d_nreq(n_reads), d_gerr(1);
	dbuf<unsigned long long> d_cnt(3);   /* SW cells, mate rescues, rescues whose forward pass was computed ahead of the decision */
	CHKA(d_regs2); CHKA(d_perr); CHKA(d_zbuf); CHKA(d_nreq); CHKA(d_cnt); CHKA(d_gerr);
	CHK(d_perr.zero()); CHK(d_cnt.zero()); CHK(d_gerr.zero());
	SSG_LAUNCH(ssg_k_copy_regs, (n_reads + block - 1) / block, block, 0, n_reads, a1.seed_off.p, a1.regs.p, a1.n_reg.p, d_r2off.p, d_regs2.p);
	const int wpb = SSG_WAVES_PER_WG;
	dbuf<unsigned int> d_q(1);
	CHKA(d_q); CHK(d_q.zero());
	if (!(opt->flag & SSG_F_NO_RESCUE)) {	/* ---- mate rescue (upstream mem_sam_pe: unless -S) ---- */
		long nwg = std::min<long>(((long)n_pairs + wpb - 1) / wpb, 256 * SSG_SW_WAVES_PER_SIMD);
		long nw = nwg * wpb;
		dbuf<uint8_t> d_tglb((size_t)nw * SSG_TWIN_GLB); dbuf<unsigned long long> d_bglb((size_t)nw * SSG_MS_BCAP); dbuf<ssg_alnreg_t> d_bcopy((size_t)nw * (128 + SSG_SDP_BIG)); dbuf<ssg_sdp_big_t> d_sdpbig((size_t)nw);
		CHKA(d_tglb); CHKA(d_bglb); CHKA(d_bcopy); CHKA(d_sdpbig);
		dbuf<int32_t> d_mtodo((size_t)n_pairs); dbuf<unsigned int> d_nmtodo(1);
		CHKA(d_mtodo); CHKA(d_nmtodo); CHK(d_nmtodo.zero());
		SSG_LAUNCH(ssg_k_matesw_need, (n_pairs + 63) / 64, 64, 0, idx->v, *opt, n_pairs, d_r2off.p, d_regs2.p, a1.n_reg.p, d_pb.p, d_pes.p, d_pw.p, d_mtodo.p, d_nmtodo.p);
		/* the windows' forward passes ahead of the decisions (k_mswlane.h): lanes per job; 0 = everything through the wave code */
		dbuf<ssg_msres_t> d_jres; dbuf<int64_t> d_jbase;
		const int ml_lanes = env_int("SSG_MSW_LANES", 4);
		bool have_slots = false;
		if (ml_lanes > 0) {
			unsigned int n_todo = 0;
			CHK(d_nmtodo.down(&n_todo, 1));
			if (n_todo > 0 && n_todo < (1u << 30)) {
				dbuf<int32_t> d_jcnt((size_t)2 * n_todo);
				if (!d_jbase.alloc((size_t)2 * n_todo + 1)) { ssg_err_msg = "device allocation failed: mate rescue slots"; return SSG_ENOMEM; }
				CHKA(d_jcnt);
				SSG_LAUNCH(ssg_k_msw_count, (n_todo + 63) / 64, 64, 0, *opt, (int)n_todo, d_mtodo.p, d_r2off.p, d_regs2.p, a1.n_reg.p, d_jcnt.p);
				int64_t nslot = 0;
				STAGE("msw_count");
				CHK(dev_exclusive_scan(d_jcnt.p, d_jbase.p, 2L * n_todo, &nslot));
				STAGE("msw_scan");
				if (nslot > 0 && nslot < (1LL << 31)) {
					dbuf<ssg_msjob_t> d_jobs((size_t)nslot); dbuf<uint64_t> d_keys((size_t)nslot); dbuf<unsigned int> d_nj(2);   /* windows; the longest */
					if (!d_jres.alloc((size_t)nslot)) { ssg_err_msg = "device allocation failed: mate rescue slots"; return SSG_ENOMEM; }
					CHKA(d_jobs); CHKA(d_keys); CHKA(d_nj); CHK(d_nj.zero()); CHK(d_jres.zero());
					SSG_LAUNCH(ssg_k_msw_emit, (nslot / 4 + 63) / 64, 64, 0, idx->v, *opt, (int)n_todo, (long)(nslot / 4), d_mtodo.p, d_jbase.p, d_off.p, d_r2off.p, d_regs2.p, a1.n_reg.p,
					           d_pb.p, d_pes.p, d_jobs.p, d_keys.p, d_nj.p);
					STAGE("msw_emit");
					int64_t seq_bytes = 0;
					CHK(rt_d2h(&seq_bytes, d_off.p + n_reads, 8));
					unsigned int njt[2] = { 0, 0 };
					CHK(d_nj.down(njt, 2));
					const unsigned int nj = njt[0];
					CHK(run_msw_lane(idx, opt, (long)nj, d_keys.p, d_jobs.p, d_seq.p, d_jres.p, (max_len + 15) / 16 * 16, (int)njt[1], ml_lanes, 0, (long)nslot, (long)seq_bytes));
					have_slots = env_int("SSG_MSW_USE", 1) != 0;   /* 0: diagnostic -- the windows are computed and not used */
					if (ssg_debug()) fprintf(stderr, "[ssgpu] mate rescue: %u listed pairs, %lld slots, %u windows ahead of the decision\n", n_todo, (long long)nslot, nj);
				}
			}
		}
		SSG_LAUNCH_W(max_len > 255, ssg_k_matesw, nwg, wpb * 64, 0, idx->v, *opt, n_pairs, d_seq.p, d_off.p, d_r2off.p, d_regs2.p, a1.n_reg.p, d_pb.p, d_pes.p,
		           d_bcopy.p, d_tglb.p, d_bglb.p, d_perr.p, d_cnt.p, d_cnt.p + 1, d_mtodo.p, d_q.p, d_sdpbig.p, d_nmtodo.p,
		           have_slots ? (const ssg_msres_t*)d_jres.p : (const ssg_msres_t*)0, have_slots ? (const int64_t*)d_jbase.p : (const int64_t*)0,
		           env_int("SSG_MSW_FIXED", 1) ? (const uint8_t*)a1.sdp_fixed.p : (const uint8_t*)0);
		CHK(rt_sync());
		if (ssg_debug()) { unsigned long long c[3]; CHK(d_cnt.down(c, 3)); fprintf(stderr, "[ssgpu] mate rescue: %llu windows aligned, %llu of them ahead of the decision\n", c[1], c[2]); }
	}
	STAGE("matesw");
	dbuf<ssg_alnreq_t> d_req((size_t)tq + 1);
	CHKA(d_req);
	{	/* ---- primary marking, pairing, MAPQ, record selection ---- */
		const int ucap = 1024;
		long nthr = std::min<long>(((long)n_pairs + 63) / 64 * 64, (long)env_int("SSG_PF_THREADS", 131072));   /* 8 waves/CU at 2 waves/SIMD (249 VGPRs); 16 KB of candidate scratch per lane */
		dbuf<ssg_pair64_t> d_v((size_t)t2 + 1), d_u((size_t)nthr * ucap);
		CHKA(d_v); CHKA(d_u);
		/* d_pw is heaviest first: pairs with long region lists get a wavefront each (k_pairw.h), the rest a lane each.  From 16 regions (64 until r06T: the lane kernel on
		 * global memory took 3.9 ms for the pairs of 7..63 regions, it takes 0.5 ms for those of 7..15, and the wave kernel 0.1 ms more: profiles/r06T_ktab_pairwave_sa_ab.json) */
		unsigned int cc[5];
		CHK(dev_class_counts(d_pkey.p, n_pairs, 0, 0, env_int("SSG_PAIR_WAVE_MIN", 16), cc));
		const int n_heavy = (int)cc[0];
		const long nwg_h = n_heavy > 0 ? std::min<long>(((long)n_heavy + wpb - 1) / wpb, (long)env_int("SSG_PFW_WGS", 512)) : 0;
		dbuf<ssg_pw_slab_t> d_slab((size_t)nwg_h * wpb + 1);
		CHKA(d_slab); CHK(d_q.zero());
		ssg_fork(1);
		if (n_heavy > 0)
			SSG_LAUNCH_ON(0, ssg_k_pair_final_wave, nwg_h, wpb * 64, 0, idx->v, *opt, n_heavy, id0, d_r2off.p, d_regs2.p, a1.n_reg.p, d_pb.p, d_pes.p, d_zbuf.p, d_v.p, d_slab.p,
			              d_reqoff.p, d_req.p, d_nreq.p, d_perr.p, d_pw.p, d_q.p);
		if (n_pairs > n_heavy) {
			/* the pairs with a handful of regions (nearly all) with their state in LDS (k_pair.h ssg_k_pair_final_lds); what it lists, on global memory.  SSG_PAIR_LDS=0: all on global memory (A/B, tests) */
			dbuf<int32_t> d_ptodo((size_t)n_pairs); dbuf<unsigned int> d_nptodo(1);
			CHKA(d_ptodo); CHKA(d_nptodo); CHK(d_nptodo.zero());
			const bool pf_lds = env_int("SSG_PAIR_LDS", 1) != 0;
			if (pf_lds) SSG_LAUNCH(ssg_k_pair_final_lds, (n_pairs - n_heavy + 63) / 64, 64, 0, idx->v, *opt, n_pairs, id0, d_r2off.p, d_regs2.p, a1.n_reg.p, d_pb.p, d_pes.p,
			                       d_reqoff.p, d_req.p, d_nreq.p, d_perr.p, d_pw.p, n_heavy, d_ptodo.p, d_nptodo.p);
			SSG_LAUNCH(ssg_k_pair_final, nthr / 64, 64, 0, idx->v, *opt, n_pairs, id0, d_r2off.p, d_regs2.p, a1.n_reg.p, d_pb.p, d_pes.p, d_zbuf.p, d_v.p, d_u.p, ucap,
			           d_reqoff.p, d_req.p, d_nreq.p, d_perr.p, d_pw.p, n_heavy, pf_lds ? (const int32_t*)d_ptodo.p : (const int32_t*)0, pf_lds ? (const unsigned int*)d_nptodo.p : (const unsigned int*)0);
			ssg_join(1);
			CHK(rt_sync());   /* (before the work lists of this scope go back to the arena) */
		} else {
			ssg_join(1);
			CHK(rt_sync());
		}
	}
	STAGE("pair_final");
	{
		unsigned int cc[5];
		CHK(dev_class_counts(d_perr.p, n_pairs, 0, 0, 0, cc));
		if (cc[4]) {
			std::vector<int32_t> perr(n_pairs);
			CHK(d_perr.down(perr.data(), n_pairs));
			for (int p = 0; p < n_pairs; ++p) if (perr[p]) { char b[128]; snprintf(b, sizeof(b), "pair %d exceeded an on-device capacity (code %d)", p, perr[p]); ssg_err_msg = b; return SSG_EOVERFLOW; }
		}
	}
	/* ---- compact the requests and generate CIGAR / NM / MD ---- */
	dbuf<int64_t> d_coff(n_reads + 1);
	CHKA(d_coff);
	int64_t nreq = 0;
	CHK(dev_exclusive_scan(d_nreq.p, d_coff.p, n_reads, &nreq));
	if (!keep) { res->req_off.resize(n_reads + 1); CHK(d_coff.down(res->req_off.data(), (size_t)n_reads + 1)); }
	dbuf<ssg_alnreq_t> d_creq((size_t)nreq + 1); dbuf<ssg_aln_t> d_alns((size_t)nreq + 1);
	CHKA(d_creq); CHKA(d_alns);
	SSG_LAUNCH(ssg_k_compact_req, (n_reads + block - 1) / block, block, 0, n_reads, d_reqoff.p, d_req.p, d_nreq.p, d_coff.p, d_creq.p);
	CHK(run_reg2aln(idx, opt, nreq, d_creq.p, d_regs2.p, d_seq.p, d_off.p, d_alns.p, d_gerr.p, d_cnt.p, max_len));
	STAGE("reg2aln");
	{ int32_t ge; CHK(d_gerr.down(&ge, 1)); if (ge) { char b[96]; snprintf(b, sizeof(b), "CIGAR generation exceeded an on-device capacity (code %d)", ge); ssg_err_msg = b; return SSG_EOVERFLOW; } }
	{ unsigned long long c[2]; CHK(d_cnt.down(c, 2)); res->stats[2] = c[0]; res->stats[3] = c[1]; res->stats[4] = (uint64_t)nreq; }
	if (keep) { keep->req.swap(d_creq); keep->alns.swap(d_alns); keep->req_off.swap(d_coff); keep->n_req = nreq; }
	else {
		if (!res->req.resize((size_t)nreq) || !res->alns.resize((size_t)nreq)) { ssg_err_msg = "host allocation failed: result records"; return SSG_ENOMEM; }
		CHK(d_creq.down(res->req.data(), (size_t)nreq)); CHK(d_alns.down(res->alns.data(), (size_t)nreq));
		STAGE("download");
	}
	return 0;
}

int ssg_pe_core(const ssg_index *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *d_seq, const int64_t *d_off, int max_len,
                const int32_t *d_pair_batch, int n_batches, int64_t id0, const ssg_pestat_t *pes0, ssg_pe_result *res, pe_dev_t *keep)
{
	return pe_core(idx, opt, n_pairs, d_seq, d_off, max_len, d_pair_batch, n_batches, id0, pes0, res, keep);
}
int ssg_dev_exclusive_scan(const int32_t *d_in, int64_t *d_out, long n, int64_t *total) { return dev_exclusive_scan(d_in, d_out, n, total); }

/* single-end reads (upstream mem_process_seqs without MEM_F_PE): stage 1 as for pairs, then every read on its own -- primary marking with
 * id = id0 + r (upstream's n_processed + i), the list of records (ssg_k_se_final), CIGAR / NM / MD.  No insert-size model, no mate rescue, no pairing. */
static int se_core(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *d_seq, const int64_t *d_off, int max_len, int64_t id0, ssg_pe_result *res)
{
	res->n_reads = n_reads; res->n_batches = 0; res->se = 1; memset(res->stats, 0, sizeof(res->stats));
	align1_dev_t a1;
	CHK(run_align1(idx, opt, n_reads, d_seq, d_off, max_len, a1, res->stats));
	const int block = 256, wpb = SSG_WAVES_PER_WG;
	dbuf<int32_t> d_capq(n_reads), d_nreq(n_reads), d_zbuf((size_t)a1.tot_seeds + 1), d_ibuf((size_t)a1.tot_seeds + 1), d_gerr(1);
	dbuf<int64_t> d_reqoff(n_reads + 1), d_coff(n_reads + 1); dbuf<unsigned long long> d_cnt(2);
	CHKA(d_capq); CHKA(d_nreq); CHKA(d_zbuf); CHKA(d_ibuf); CHKA(d_gerr); CHKA(d_reqoff); CHKA(d_coff); CHKA(d_cnt);
	CHK(d_gerr.zero()); CHK(d_cnt.zero());
	SSG_LAUNCH(ssg_k_se_caps, (n_reads + block - 1) / block, block, 0, n_reads, a1.n_reg.p, d_capq.p);
	int64_t tq = 0;
	CHK(dev_exclusive_scan(d_capq.p, d_reqoff.p, n_reads, &tq));
	dbuf<ssg_alnreq_t> d_req((size_t)tq + 1);
	CHKA(d_req);
	SSG_LAUNCH(ssg_k_se_final, (n_reads + 63) / 64, 64, 0, *opt, n_reads, id0, a1.seed_off.p, a1.regs.p, a1.n_reg.p, d_zbuf.p, d_ibuf.p, d_reqoff.p, d_req.p, d_nreq.p);
	STAGE("se_final");
	int64_t nreq = 0;
	CHK(dev_exclusive_scan(d_nreq.p, d_coff.p, n_reads, &nreq));
	res->req_off.resize((size_t)n_reads + 1); CHK(d_coff.down(res->req_off.data(), (size_t)n_reads + 1));
	dbuf<ssg_alnreq_t> d_creq((size_t)nreq + 1); dbuf<ssg_aln_t> d_alns((size_t)nreq + 1);
	CHKA(d_creq); CHKA(d_alns);
	SSG_LAUNCH(ssg_k_compact_req, (n_reads + block - 1) / block, block, 0, n_reads, d_reqoff.p, d_req.p, d_nreq.p, d_coff.p, d_creq.p);
	CHK(run_reg2aln(idx, opt, nreq, d_creq.p, a1.regs.p, d_seq, d_off, d_alns.p, d_gerr.p, d_cnt.p, max_len));
	STAGE("reg2aln");
	{ int32_t ge; CHK(d_gerr.down(&ge, 1)); if (ge) { char b[96]; snprintf(b, sizeof(b), "CIGAR generation exceeded an on-device capacity (code %d)", ge); ssg_err_msg = b; return SSG_EOVERFLOW; } }
	{ unsigned long long c[2]; CHK(d_cnt.down(c, 2)); res->stats[2] = c[0]; res->stats[3] = c[1]; res->stats[4] = (uint64_t)nreq; }
	if (!res->req.resize((size_t)nreq) || !res->alns.resize((size_t)nreq)) { ssg_err_msg = "host allocation failed: result records"; return SSG_ENOMEM; }
	CHK(d_creq.down(res->req.data(), (size_t)nreq)); CHK(d_alns.down(res->alns.data(), (size_t)nreq));
	STAGE("download");
	return 0;
}

/* stable sort of (hash, ordinal) by hash: hipCUB radix sort on the GPU */
static int sort_pairs_u64(uint64_t *k_in, uint64_t *k_out, uint32_t *v_in, uint32_t *v_out, long n)
{
#ifdef SSG_EMU
	std::vector<uint32_t> idx(n); for (long i = 0; i < n; ++i) idx[i] = (uint32_t)i;
	std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return k_in[a] < k_in[b]; });
	for (long i = 0; i < n; ++i) { k_out[i] = k_in[idx[i]]; v_out[i] = v_in[idx[i]]; }
	return 0;
#else
	size_t tmp_bytes = 0;
	if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, 64u) != hipSuccess) { ssg_err_msg = "rocprim radix_sort_pairs (size query) failed"; return SSG_EHIP; }
	dbuf<uint8_t> tmp(tmp_bytes);
	CHKA(tmp);
	if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, 64u, ssg_stream) != hipSuccess) { ssg_err_msg = "rocprim radix_sort_pairs failed"; return SSG_EHIP; }
	return rt_sync();
#endif
}

/* ---- the persistent duplicate set (k_sbl.h): open-addressing table in HBM ---- */
struct ssg_sbl_state {
	uint64_t *th; ssg_sig_t *ts; uint64_t slots, n;     /* slots = power of two (0 = no table yet), n = signatures held */
	ssg_sbl_state() : th(0), ts(0), slots(0), n(0) {}
	~ssg_sbl_state() { rt_free_raw(th); rt_free_raw(ts); }
};
static int sbl_table_reserve(ssg_sbl_state *st, uint64_t more)
{	/* load factor <= 1/2 after `more` insertions */
	uint64_t want = st->slots ? st->slots : (uint64_t)std::max(16, env_int("SSG_SBL_TABLE_SLOTS", 1 << 20));   /* power of two; the tests start small to exercise the growth */
	while ((st->n + more) * 2 > want) want <<= 1;
	if (want == st->slots) return 0;
	uint64_t *nh = (uint64_t*)rt_malloc_raw(want * 8); ssg_sig_t *ns = (ssg_sig_t*)rt_malloc_raw(want * sizeof(ssg_sig_t));
	if (!nh || !ns) { rt_free_raw(nh); rt_free_raw(ns); ssg_err_msg = "device allocation failed: duplicate-signature table"; return SSG_ENOMEM; }
	CHK(rt_memset(nh, 0, want * 8));
	if (st->slots) SSG_LAUNCH(ssg_k_sbl_rehash, (st->slots + 255) / 256, 256, 0, st->slots, st->th, st->ts, nh, ns, want - 1);
	CHK(rt_sync());
	rt_free_raw(st->th); rt_free_raw(st->ts);
	st->th = nh; st->ts = ns; st->slots = want;
	return 0;
}
/* dup[p] for the pairs whose ends are in d_ends (device), first-seen-wins by position in the call; with a state, signatures seen in
 * earlier calls count and this call's new signatures are added (upstream's single pass over the whole stream) */
static int dedup_core(ssg_sbl_state *st, long n_pairs, const ssg_sbl_end_t *d_ends, uint8_t *d_dup, ssg_sig_t *d_sig_out = 0)
{
	const int block = 256;
	dbuf<ssg_sig_t> d_sigb; dbuf<uint64_t> d_h(n_pairs), d_hs(n_pairs); dbuf<uint32_t> d_o(n_pairs), d_os(n_pairs); dbuf<uint8_t> d_fresh;
	if (!d_sig_out) { if (!d_sigb.alloc(n_pairs)) { ssg_err_msg = "device allocation failed: signatures"; return SSG_ENOMEM; } d_sig_out = d_sigb.p; }
	CHKA(d_h); CHKA(d_hs); CHKA(d_o); CHKA(d_os);
	if (st) { if (!d_fresh.alloc(n_pairs)) { ssg_err_msg = "device allocation failed: fresh flags"; return SSG_ENOMEM; } CHK(sbl_table_reserve(st, (uint64_t)n_pairs)); }
	SSG_LAUNCH(ssg_k_sig, (n_pairs + block - 1) / block, block, 0, n_pairs, d_ends, d_sig_out, d_h.p, d_o.p);
	CHK(sort_pairs_u64(d_h.p, d_hs.p, d_o.p, d_os.p, n_pairs));
	SSG_LAUNCH(ssg_k_sbl_markdup, (n_pairs + block - 1) / block, block, 0, n_pairs, d_hs.p, d_os.p, d_sig_out,
	           st ? st->th : (const uint64_t*)0, st ? st->ts : (const ssg_sig_t*)0, st ? st->slots - 1 : 0, d_dup, st ? d_fresh.p : (uint8_t*)0);
	if (st) {
		SSG_LAUNCH(ssg_k_sbl_insert, (n_pairs + block - 1) / block, block, 0, n_pairs, d_h.p, d_sig_out, d_fresh.p, st->th, st->ts, st->slots - 1);
		st->n += (uint64_t)n_pairs;   /* upper bound (duplicates and never-duplicate pairs are not inserted): only steers the growth */
	}
	return rt_sync();
}

__global__ void ssg_k_iota_u32(uint32_t *a, long n) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = (uint32_t)i; }
/* first-seen-wins over signatures that arrive from several ranks (SURVEY 8e coupling 2, the owner's side of the exchange): element i
 * is a duplicate iff another element with the same signature has a smaller ordinal.  All-ones signatures never are.  One radix
 * sort by ordinal puts the elements in input order; from there it is the single-process duplicate marking (hash sort + run scan). */
__global__ void ssg_k_sigdup_gather(long n, const uint32_t *perm, const ssg_sig_t *sig, ssg_sig_t *sig_o, uint64_t *hash, uint32_t *ord)
{
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const ssg_sig_t s = sig[perm[i]];
	sig_o[i] = s; ord[i] = (uint32_t)i;
	hash[i] = ssg_sig_never(s) ? ~0ull - (uint64_t)i : ssg_mix64(s.k0 ^ ssg_mix64(s.k1 ^ ssg_mix64(s.k2)));
}
__global__ void ssg_k_sigdup_scatter(long n, const uint32_t *perm, const uint8_t *dup_o, uint8_t *dup)
{
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dup[perm[i]] = dup_o[i];
}
static int sigdup_core(long n, const ssg_sig_t *d_sig, const uint64_t *d_ordinal, uint8_t *d_dup)
{
	const int block = 256;
	dbuf<uint64_t> d_k(n), d_ks(n), d_h(n), d_hs(n); dbuf<uint32_t> d_v(n), d_perm(n), d_o(n), d_os(n); dbuf<ssg_sig_t> d_so(n); dbuf<uint8_t> d_do(n);
	CHKA(d_k); CHKA(d_ks); CHKA(d_h); CHKA(d_hs); CHKA(d_v); CHKA(d_perm); CHKA(d_o); CHKA(d_os); CHKA(d_so); CHKA(d_do);
	CHK(rt_d2d(d_k.p, d_ordinal, (size_t)n * 8));
	SSG_LAUNCH(ssg_k_iota_u32, (n + block - 1) / block, block, 0, d_v.p, n);
	CHK(sort_pairs_u64(d_k.p, d_ks.p, d_v.p, d_perm.p, n));
	SSG_LAUNCH(ssg_k_sigdup_gather, (n + block - 1) / block, block, 0, n, d_perm.p, d_sig, d_so.p, d_h.p, d_o.p);
	CHK(sort_pairs_u64(d_h.p, d_hs.p, d_o.p, d_os.p, n));
	SSG_LAUNCH(ssg_k_sbl_markdup, (n + block - 1) / block, block, 0, n, d_hs.p, d_os.p, d_so.p, (const uint64_t*)0, (const ssg_sig_t*)0, (uint64_t)0, d_do.p, (uint8_t*)0);
	SSG_LAUNCH(ssg_k_sigdup_scatter, (n + block - 1) / block, block, 0, n, d_perm.p, d_do.p, d_dup);
	return rt_sync();
}

/* rows a14-a17 for a chunk of name-grouped blocks resident in HBM: duplicate bits (through the state's table), mate lines, side-stream bits */
static int sbl_process_dev(ssg_sbl_state *st, const ssg_sbl_opt_t *o, long n_blocks, const int64_t *d_blk_off, const ssg_sbl_line_t *d_lines,
                           uint8_t *d_out, int64_t *d_mate, uint8_t *d_dup_out /* may be NULL */)
{
	const int block = 256;
	if (o->max_split_count > SSG_SBL_MAX_SPLIT) { ssg_err_msg = "samblaster: --maxSplitCount above 16 is not supported"; return SSG_EINVAL; }
	dbuf<ssg_sbl_end_t> d_ends(2 * n_blocks); dbuf<int64_t> d_prim(2 * n_blocks); dbuf<uint8_t> d_dupb;
	CHKA(d_ends); CHKA(d_prim);
	if (!d_dup_out) { if (!d_dupb.alloc(n_blocks)) { ssg_err_msg = "device allocation failed: dup flags"; return SSG_ENOMEM; } d_dup_out = d_dupb.p; }
	SSG_LAUNCH(ssg_k_sbl_ends, (n_blocks + block - 1) / block, block, 0, n_blocks, d_blk_off, d_lines, d_ends.p, d_prim.p);
	CHK(dedup_core(st, n_blocks, d_ends.p, d_dup_out));
	SSG_LAUNCH(ssg_k_sbl_classify, (n_blocks + block - 1) / block, block, 0, *o, n_blocks, d_blk_off, d_lines, d_prim.p, d_dup_out, d_out, d_mate);
	return rt_sync();
}

extern "C" {

static int process_pairs_host(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *seq, const int64_t *off,
                              const int32_t *pair_batch, int n_batches, int64_t id0, const ssg_pestat_t *pes0, ssg_pe_result_t **out);

int ssg_mem_process_pairs(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *seq, const int64_t *off,
                          const int32_t *pair_batch, int n_batches, int64_t id0, const ssg_pestat_t *pes0, ssg_pe_result_t **out)
{
	return process_pairs_host(idx, opt, n_pairs, seq, off, pair_batch, n_batches, id0, pes0, out);
}
int ssg_mem_process_reads(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *seq, const int64_t *off, int64_t id0, ssg_pe_result_t **out)
{
	CHK(need_device());
	*out = 0;
	if (n_reads <= 0) { ssg_err_msg = "ssg_mem_process_reads: empty input"; return SSG_EINVAL; }
	int max_len = 0; for (int r = 0; r < n_reads; ++r) max_len = std::max<int>(max_len, (int)(off[r+1] - off[r]));
	if (max_len > SSG_MAX_READ_LEN) { ssg_err_msg = "reads longer than " SSG_STR(SSG_MAX_READ_LEN) " bases are outside this build's scope"; return SSG_EINVAL; }
	dbuf<uint8_t> d_seq((size_t)off[n_reads] + 1); dbuf<int64_t> d_off(n_reads + 1);
	CHKA(d_seq); CHKA(d_off);
	if (ssg_debug()) (void)ssg_stage_ms();
	CHK(d_seq.up(seq, off[n_reads])); CHK(d_off.up(off, n_reads + 1));
	STAGE("upload");
	std::unique_ptr<ssg_pe_result> res(new ssg_pe_result());
	CHK(se_core(idx, opt, n_reads, d_seq.p, d_off.p, max_len, id0, res.get()));
	ssg_prof_flush();
	*out = res.release();
	return 0;
}
/* upstream samblaster duplicate marking (row a14) on per-end records supplied by the caller
 * (2*n_pairs entries: read1, read2 primaries); dup[p] = 1 when an earlier pair has the same signature */
int ssg_sbl_markdup(long n_pairs, const ssg_sbl_end_t *ends, uint8_t *dup)
{
	CHK(need_device());
	if (n_pairs <= 0) return 0;
	if (n_pairs >= (1L << 31)) { ssg_err_msg = "ssg_sbl_markdup: more than 2^31 pairs per call"; return SSG_EINVAL; }
	dbuf<ssg_sbl_end_t> d_ends(2 * n_pairs); dbuf<uint8_t> d_dup(n_pairs);
	CHKA(d_ends); CHKA(d_dup);
	CHK(d_ends.up(ends, 2 * n_pairs));
	CHK(dedup_core(0, n_pairs, d_ends.p, d_dup.p));
	return d_dup.down(dup, n_pairs);
}

/* Streaming duplicate marking: the signatures of everything seen so far stay in an HBM hash table (k_sbl.h), so first-seen-wins
 * holds across calls exactly as in upstream's single pass over the stream; each call costs O(its own pairs). */
ssg_sbl_state_t *ssg_sbl_state_new(void) { return new ssg_sbl_state(); }
void ssg_sbl_state_free(ssg_sbl_state_t *s) { delete s; }

int ssg_sbl_markdup_stream(ssg_sbl_state_t *st, long n_pairs, const ssg_sbl_end_t *ends, uint8_t *dup)
{
	CHK(need_device());
	if (n_pairs <= 0) return 0;
	if (n_pairs >= (1L << 31)) { ssg_err_msg = "ssg_sbl_markdup_stream: more than 2^31 pairs per call"; return SSG_EINVAL; }
	dbuf<ssg_sbl_end_t> d_ends(2 * n_pairs); dbuf<uint8_t> d_dup(n_pairs);
	CHKA(d_ends); CHKA(d_dup);
	CHK(d_ends.up(ends, 2 * n_pairs));
	CHK(dedup_core(st, n_pairs, d_ends.p, d_dup.p));
	return d_dup.down(dup, n_pairs);
}

/* the owner's side of the duplicate exchange between ranks (speedseq_amd/dist.py global_markdup): DEVICE pointers; d_sig n x 3 uint64 as
 * ssg_hotpath_dev_ex wrote them on the sending ranks (all ones = never a duplicate), d_ordinal the global input ordinal of each pair */
int ssg_markdup_sig_dev(long n, const uint64_t *d_sig, const int64_t *d_ordinal, uint8_t *d_dup)
{
	CHK(need_device());
	if (n <= 0) return 0;
	if (n >= (1L << 32)) { ssg_err_msg = "ssg_markdup_sig_dev: more than 2^32 signatures per call"; return SSG_EINVAL; }
	return sigdup_core(n, (const ssg_sig_t*)d_sig, (const uint64_t*)d_ordinal, d_dup);
}

/* stable sort of 64-bit keys on the device: perm[i] = input index of the i-th smallest key (equal keys keep input order).
 * The coordinate sort of BAM records (samtools bam_sort.c:1607-1614 key, stable) is this call on the record keys (row f1). */
int ssg_sort_u64_perm(const uint64_t *keys, int64_t n, uint32_t *perm)
{
	CHK(need_device());
	if (n <= 0) return 0;
	if (n >= (1LL << 32)) { ssg_err_msg = "ssg_sort_u64_perm: more than 2^32 keys per call"; return SSG_EINVAL; }
	dbuf<uint64_t> d_k(n), d_ks(n); dbuf<uint32_t> d_v(n), d_vs(n);
	CHKA(d_k); CHKA(d_ks); CHKA(d_v); CHKA(d_vs);
	CHK(d_k.up(keys, n));
	SSG_LAUNCH(ssg_k_iota_u32, (n + 255) / 256, 256, 0, d_v.p, (long)n);
	CHK(sort_pairs_u64(d_k.p, d_ks.p, d_v.p, d_vs.p, n));
	return d_vs.down(perm, n);
}

void ssg_sbl_opt_init(ssg_sbl_opt_t *o)
{	/* upstream samblaster defaults */
	o->exclude_dups = 0; o->add_mate_tags = 0; o->max_split_count = 2; o->min_non_overlap = 20; o->max_unmapped_bases = 50; o->min_indel_size = 50;
}

/* upstream samblaster's per-block decisions (rows a14-a17) for a chunk of the stream; host buffers */
int ssg_sbl_process(ssg_sbl_state_t *st, const ssg_sbl_opt_t *o, long n_blocks, const int64_t *blk_off, const ssg_sbl_line_t *lines,
                    uint8_t *line_bits, int64_t *mate_line)
{
	CHK(need_device());
	if (n_blocks <= 0) return 0;
	if (n_blocks >= (1L << 31)) { ssg_err_msg = "ssg_sbl_process: more than 2^31 blocks per call"; return SSG_EINVAL; }
	const int64_t n_lines = blk_off[n_blocks];
	dbuf<int64_t> d_off(n_blocks + 1), d_mate(n_lines + 1); dbuf<ssg_sbl_line_t> d_lines(n_lines + 1); dbuf<uint8_t> d_out(n_lines + 1);
	CHKA(d_off); CHKA(d_mate); CHKA(d_lines); CHKA(d_out);
	CHK(d_off.up(blk_off, n_blocks + 1)); CHK(d_lines.up(lines, n_lines));
	CHK(sbl_process_dev(st, o, n_blocks, d_off.p, d_lines.p, d_out.p, d_mate.p, 0));
	CHK(d_out.down(line_bits, n_lines));
	return d_mate.down(mate_line, n_lines);
}

/* ssg_sbl_process in two halves, for pipelines that run side by side and share ONE duplicate set (rank mode): the ends of
 * every block's primaries (what the signature is built from) go to the process that owns the set (ssg_sbl_markdup_stream there, batches in
 * input order), its verdicts come back, and the lines are classified with them.  Same kernels as ssg_sbl_process, same results. */
int ssg_sbl_ends(long n_blocks, const int64_t *blk_off, const ssg_sbl_line_t *lines, ssg_sbl_end_t *ends)
{
	CHK(need_device());
	if (n_blocks <= 0) return 0;
	if (n_blocks >= (1L << 31)) { ssg_err_msg = "ssg_sbl_ends: more than 2^31 blocks per call"; return SSG_EINVAL; }
	const int64_t n_lines = blk_off[n_blocks]; const int block = 256;
	dbuf<int64_t> d_off(n_blocks + 1), d_prim(2 * n_blocks); dbuf<ssg_sbl_line_t> d_lines(n_lines + 1); dbuf<ssg_sbl_end_t> d_ends(2 * n_blocks);
	CHKA(d_off); CHKA(d_prim); CHKA(d_lines); CHKA(d_ends);
	CHK(d_off.up(blk_off, n_blocks + 1)); CHK(d_lines.up(lines, n_lines));
	SSG_LAUNCH(ssg_k_sbl_ends, (n_blocks + block - 1) / block, block, 0, n_blocks, d_off.p, d_lines.p, d_ends.p, d_prim.p);
	return d_ends.down(ends, 2 * n_blocks);
}
int ssg_sbl_classify(const ssg_sbl_opt_t *o, long n_blocks, const int64_t *blk_off, const ssg_sbl_line_t *lines, const uint8_t *dup, uint8_t *line_bits, int64_t *mate_line)
{
	CHK(need_device());
	if (n_blocks <= 0) return 0;
	if (n_blocks >= (1L << 31)) { ssg_err_msg = "ssg_sbl_classify: more than 2^31 blocks per call"; return SSG_EINVAL; }
	if (o->max_split_count > SSG_SBL_MAX_SPLIT) { ssg_err_msg = "samblaster: --maxSplitCount above 16 is not supported"; return SSG_EINVAL; }
	const int64_t n_lines = blk_off[n_blocks]; const int block = 256;
	dbuf<int64_t> d_off(n_blocks + 1), d_prim(2 * n_blocks), d_mate(n_lines + 1); dbuf<ssg_sbl_line_t> d_lines(n_lines + 1); dbuf<ssg_sbl_end_t> d_ends(2 * n_blocks);
	dbuf<uint8_t> d_dup(n_blocks), d_out(n_lines + 1);
	CHKA(d_off); CHKA(d_prim); CHKA(d_mate); CHKA(d_lines); CHKA(d_ends); CHKA(d_dup); CHKA(d_out);
	CHK(d_off.up(blk_off, n_blocks + 1)); CHK(d_lines.up(lines, n_lines)); CHK(d_dup.up(dup, n_blocks));
	SSG_LAUNCH(ssg_k_sbl_ends, (n_blocks + block - 1) / block, block, 0, n_blocks, d_off.p, d_lines.p, d_ends.p, d_prim.p);
	SSG_LAUNCH(ssg_k_sbl_classify, (n_blocks + block - 1) / block, block, 0, *o, n_blocks, d_off.p, d_lines.p, d_prim.p, d_dup.p, d_out.p, d_mate.p);
	CHK(rt_sync());
	CHK(d_out.down(line_bits, n_lines));
	return d_mate.down(mate_line, n_lines);
}

/* aligned records of one call kept in HBM with their samblaster view (lines, primaries), for the stages after alignment */
struct ssg_dev_records {
	pe_dev_t keep; long n_pairs; int64_t n_lines;
	dbuf<int64_t> line_off, line_req, prim, mate; dbuf<ssg_sbl_line_t> lines; dbuf<ssg_sbl_end_t> ends; dbuf<uint8_t> bits, dup;
};

/* SAM lines of the kept records (k_sbl.h) and the two primary ends per pair */
static int records_lines(struct ssg_dev_records *R)
{
	const long n_pairs = R->n_pairs; const int block = 256;
	dbuf<int32_t> d_nl(n_pairs);
	CHKA(d_nl);
	if (!R->line_off.alloc(n_pairs + 1)) { ssg_err_msg = "device allocation failed: line offsets"; return SSG_ENOMEM; }
	SSG_LAUNCH(ssg_k_sbl_count_lines, (n_pairs + block - 1) / block, block, 0, n_pairs, R->keep.req_off.p, R->keep.req.p, d_nl.p);
	CHK(dev_exclusive_scan(d_nl.p, R->line_off.p, n_pairs, &R->n_lines));
	const size_t nl = (size_t)R->n_lines + 1;
	if (!R->lines.alloc(nl) || !R->line_req.alloc(nl) || !R->prim.alloc(2 * n_pairs) || !R->mate.alloc(nl) || !R->bits.alloc(nl) || !R->ends.alloc(2 * n_pairs) || !R->dup.alloc(n_pairs)) {
		ssg_err_msg = "device allocation failed: samblaster lines"; return SSG_ENOMEM; }
	SSG_LAUNCH(ssg_k_sbl_lines_from_alns, (n_pairs + block - 1) / block, block, 0, n_pairs, R->keep.req_off.p, R->keep.req.p, R->keep.alns.p, R->line_off.p, R->lines.p, R->line_req.p);
	SSG_LAUNCH(ssg_k_sbl_ends, (n_pairs + block - 1) / block, block, 0, n_pairs, R->line_off.p, R->lines.p, R->ends.p, R->prim.p);
	return 0;
}
static int process_pairs_host(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *seq, const int64_t *off,
                              const int32_t *pair_batch, int n_batches, int64_t id0, const ssg_pestat_t *pes0, ssg_pe_result_t **out)
{
	CHK(need_device());
	*out = 0;
	const int n_reads = 2 * n_pairs;
	if (n_pairs <= 0 || n_batches <= 0) { ssg_err_msg = "ssg_mem_process_pairs: empty input"; return SSG_EINVAL; }
	int max_len = 0; for (int r = 0; r < n_reads; ++r) max_len = std::max<int>(max_len, (int)(off[r+1] - off[r]));
	if (max_len > SSG_MAX_READ_LEN) { ssg_err_msg = "reads longer than " SSG_STR(SSG_MAX_READ_LEN) " bases are outside this build's scope"; return SSG_EINVAL; }
	for (int p = 0; p < n_pairs; ++p) if (pair_batch[p] < 0 || pair_batch[p] >= n_batches) { ssg_err_msg = "pair_batch out of range"; return SSG_EINVAL; }
	dbuf<uint8_t> d_seq((size_t)off[n_reads] + 1); dbuf<int64_t> d_off(n_reads + 1); dbuf<int32_t> d_pb(n_pairs);
	CHKA(d_seq); CHKA(d_off); CHKA(d_pb);
	if (ssg_debug()) (void)ssg_stage_ms();
	CHK(d_seq.up(seq, off[n_reads])); CHK(d_off.up(off, n_reads + 1)); CHK(d_pb.up(pair_batch, n_pairs));
	STAGE("upload");
	std::unique_ptr<ssg_pe_result> res(new ssg_pe_result());
	CHK(pe_core(idx, opt, n_pairs, d_seq.p, d_off.p, max_len, d_pb.p, n_batches, id0, pes0, res.get(), 0));
	ssg_prof_flush();
	*out = res.release();
	return 0;
}
static int records_classify(ssg_dev_records *R, const ssg_sbl_opt_t *o, const uint8_t *d_dup, uint64_t counts[4])
{	/* counts: [0] duplicate pairs [1] lines to the discordant stream [2] lines to the splitter stream [3] SAM lines */
	const long n_pairs = R->n_pairs; const int block = 256;
	if (o->max_split_count > SSG_SBL_MAX_SPLIT) { ssg_err_msg = "samblaster: --maxSplitCount above 16 is not supported"; return SSG_EINVAL; }
	SSG_LAUNCH(ssg_k_sbl_classify, (n_pairs + block - 1) / block, block, 0, *o, n_pairs, R->line_off.p, R->lines.p, R->prim.p, d_dup, R->bits.p, R->mate.p);
	dbuf<unsigned int> d_c(4);
	CHKA(d_c); CHK(d_c.zero());
	SSG_LAUNCH(ssg_k_sbl_count_bits, (R->n_lines + block - 1) / block, block, 0, R->n_lines, R->bits.p, n_pairs, d_dup, d_c.p);
	unsigned int c[4];
	CHK(d_c.down(c, 4));
	counts[0] = c[0]; counts[1] = c[1]; counts[2] = c[2]; counts[3] = (uint64_t)R->n_lines;
	return 0;
}

/* The measured hot path (bench.py): device-resident reads in; aligned, duplicate-marked and side-stream-classified records
 * left in HBM (rows a1-a12, a14-a17).  d_seq / d_off / d_pair_batch are DEVICE pointers.  summary[0] = records,
 * [1] = duplicate pairs, [2] = seeds, [3] = extension cells, [4] = rescue cells, [5] = rescues, [6] = bwt_extend calls, [7] = chains,
 * [8] = discordant-stream lines, [9] = splitter-stream lines, [10] = SAM lines.
 * local_dedup = 0 leaves duplicate marking and classification to the caller (sharded input: signatures in d_sig_out are
 * exchanged between ranks first, then ssg_dev_records_classify runs on the global verdicts). */
int ssg_hotpath_dev_ex(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, int max_len, const uint8_t *d_seq, const int64_t *d_off,
                       const int32_t *d_pair_batch, int n_batches, int64_t id0, const ssg_sbl_opt_t *sbl, int local_dedup,
                       uint64_t summary[16], uint8_t *dup_host, uint64_t *d_sig_out, ssg_dev_records_t **keep_out)
{
	CHK(need_device());
	if (keep_out) *keep_out = 0;
	if (n_pairs <= 0 || max_len > SSG_MAX_READ_LEN) { ssg_err_msg = "ssg_hotpath_dev: bad arguments"; return SSG_EINVAL; }
	ssg_sbl_opt_t so; if (sbl) so = *sbl; else { ssg_sbl_opt_init(&so); so.exclude_dups = 1; so.add_mate_tags = 1; }   /* the reference's command line */
	ssg_pe_result res; std::unique_ptr<ssg_dev_records> R(new ssg_dev_records());
	R->n_pairs = n_pairs;
	CHK(pe_core(idx, opt, n_pairs, d_seq, d_off, max_len, d_pair_batch, n_batches, id0, 0, &res, &R->keep));
	CHK(records_lines(R.get()));
	memset(summary, 0, 16 * sizeof(uint64_t));
	if (local_dedup || d_sig_out) {
		dbuf<ssg_sig_t> d_sigb;
		CHK(dedup_core(0, n_pairs, R->ends.p, R->dup.p, (ssg_sig_t*)d_sig_out));
	}
	if (local_dedup) {
		uint64_t c[4];
		CHK(records_classify(R.get(), &so, R->dup.p, c));
		summary[1] = c[0]; summary[8] = c[1]; summary[9] = c[2]; summary[10] = c[3];
		if (dup_host) CHK(R->dup.down(dup_host, n_pairs));
	}
	summary[0] = res.stats[4]; summary[2] = res.stats[0]; summary[3] = res.stats[1]; summary[4] = res.stats[2]; summary[5] = res.stats[3];
	summary[7] = res.stats[6]; /* chains = first-seed extensions */
	summary[11] = res.stats[7]; /* reads chained again on klib's B-tree (more than 9 chains and two at one position) */
	summary[6] = res.stats[5]; /* bwt_extend calls in the SMEM kernel (2 rank queries = 2 x 64-byte lines each) */
	if (keep_out) *keep_out = R.release();
	return 0;
}
int ssg_dev_records_classify(ssg_dev_records_t *R, const ssg_sbl_opt_t *sbl, const uint8_t *d_dup, uint64_t counts[4])
{
	ssg_sbl_opt_t so; if (sbl) so = *sbl; else { ssg_sbl_opt_init(&so); so.exclude_dups = 1; so.add_mate_tags = 1; }
	return records_classify(R, &so, d_dup, counts);
}
void ssg_dev_records_free(ssg_dev_records_t *R) { delete R; }
int64_t ssg_dev_records_n_lines(const ssg_dev_records_t *R) { return R->n_lines; }
size_t ssg_dev_record_bytes(void) { return sizeof(ssg_aln_t); }
/* keys (n_lines x u64), records (n_lines x ssg_dev_record_bytes(), may be NULL) and side-stream bits (n_lines, may be NULL) into DEVICE buffers of the caller */
int ssg_dev_records_export(const ssg_dev_records_t *R, uint64_t *d_keys, void *d_recs, uint8_t *d_bits)
{
	if (R->n_lines > 0) SSG_LAUNCH(ssg_k_sbl_export, (R->n_lines + 255) / 256, 256, 0, R->n_lines, R->lines.p, R->line_req.p, R->keep.alns.p, R->bits.p, d_keys, (ssg_aln_t*)d_recs, d_bits);
	return rt_sync();
}

/* the kept records back on the host as an ssg_pe_result_t (ssg_pe_* accessors, ssg_sam_format) together with, per SAM line in line
 * order, the side-stream bits and the line whose CIGAR / MAPQ fill MC / MQ: what a checker needs to compare the device step
 * record by record with upstream's `bwa mem | samblaster` output (bench.py's parity gate on the timed call, tests) */
int ssg_dev_records_download(const ssg_dev_records_t *R, ssg_pe_result_t **out, uint8_t *line_bits, int64_t *mate_line)
{
	CHK(need_device());
	*out = 0;
	std::unique_ptr<ssg_pe_result> res(new ssg_pe_result());
	const int n_reads = (int)(2 * R->n_pairs); const size_t nreq = (size_t)R->keep.n_req;
	res->n_reads = n_reads; res->n_batches = 0; memset(res->stats, 0, sizeof(res->stats));
	res->req_off.resize((size_t)n_reads + 1);
	if (!res->req.resize(nreq) || !res->alns.resize(nreq)) { ssg_err_msg = "host allocation failed: result records"; return SSG_ENOMEM; }
	CHK(rt_sync());
	CHK(R->keep.req_off.down(res->req_off.data(), (size_t)n_reads + 1)); CHK(R->keep.req.down(res->req.data(), nreq)); CHK(R->keep.alns.down(res->alns.data(), nreq));
	if (line_bits) CHK(R->bits.down(line_bits, (size_t)R->n_lines));
	if (mate_line) CHK(R->mate.down(mate_line, (size_t)R->n_lines));
	*out = res.release();
	return 0;
}

int ssg_hotpath_dev(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, int max_len, const uint8_t *d_seq, const int64_t *d_off,
                    const int32_t *d_pair_batch, int n_batches, int64_t id0, uint64_t summary[8], uint8_t *dup_host /* may be NULL */)
{
	uint64_t s16[16];
	CHK(ssg_hotpath_dev_ex(idx, opt, n_pairs, max_len, d_seq, d_off, d_pair_batch, n_batches, id0, 0, 1, s16, dup_host, 0, 0));
	memcpy(summary, s16, 8 * sizeof(uint64_t));
	return 0;
}

/* same, and the pair signatures (n_pairs x 3 uint64, DEVICE memory) for exact duplicate marking across ranks */
int ssg_hotpath_dev_sig(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, int max_len, const uint8_t *d_seq, const int64_t *d_off,
                        const int32_t *d_pair_batch, int n_batches, int64_t id0, uint64_t summary[8], uint8_t *dup_host, uint64_t *d_sig_out)
{
	uint64_t s16[16];
	CHK(ssg_hotpath_dev_ex(idx, opt, n_pairs, max_len, d_seq, d_off, d_pair_batch, n_batches, id0, 0, 1, s16, dup_host, d_sig_out, 0));
	memcpy(summary, s16, 8 * sizeof(uint64_t));
	return 0;
}

int ssg_index_set_names(ssg_index_t *ix, int n, const char *const *names)
{
	if (n != ix->v.n_ctg) { ssg_err_msg = "ssg_index_set_names: contig count mismatch"; return SSG_EINVAL; }
	ix->names.assign(names, names + n);
	{ std::lock_guard<std::mutex> l(ix->names_mu); rt_free(ix->d_names); rt_free(ix->d_name_off); ix->d_names = 0; ix->d_name_off = 0; }
	return 0;
}
const char *ssg_index_name(const ssg_index_t *ix, int i) { return i >= 0 && i < (int)ix->names.size() ? ix->names[i].c_str() : "*"; }
int32_t ssg_index_len(const ssg_index_t *ix, int i) { return i >= 0 && i < (int)ix->h_len.size() ? ix->h_len[i] : 0; }

void ssg_pe_result_free(ssg_pe_result_t *r) { delete r; }
int ssg_pe_reserve(int n_pairs, int n_calls)
{
	CHK(need_device());
	if (n_pairs <= 0 || n_calls <= 0 || n_calls > 8) { ssg_err_msg = "ssg_pe_reserve: bad arguments"; return SSG_EINVAL; }
	const size_t nrec = (size_t)n_pairs * 2 + (size_t)n_pairs / 4;      /* records per pair: 2 + supplementary / XA entries */
	std::vector<void*> a, b;
	for (int i = 0; i < n_calls; ++i) { a.push_back(rt_host_alloc(nrec * sizeof(ssg_aln_t))); b.push_back(rt_host_alloc(nrec * sizeof(ssg_alnreq_t))); }
	int rc = 0;
	for (void *p : a) { if (!p) rc = SSG_ENOMEM; rt_host_free(p); }
	for (void *p : b) { if (!p) rc = SSG_ENOMEM; rt_host_free(p); }
	if (rc) ssg_err_msg = "host allocation failed: page-locked result blocks";
	return rc;
}
int64_t ssg_pe_n_req(const ssg_pe_result_t *r) { return (int64_t)r->req.size(); }
const int64_t *ssg_pe_req_off(const ssg_pe_result_t *r) { return r->req_off.data(); }
int ssg_pe_is_se(const ssg_pe_result_t *r) { return r->se; }
const ssg_alnreq_t *ssg_pe_req(const ssg_pe_result_t *r) { return r->req.data(); }
const ssg_aln_t *ssg_pe_alns(const ssg_pe_result_t *r) { return r->alns.data(); }
const ssg_pestat_t *ssg_pe_pes(const ssg_pe_result_t *r) { return r->pes.data(); }
const uint64_t *ssg_pe_stats(const ssg_pe_result_t *r) { return r->stats; }

} /* extern "C" */
