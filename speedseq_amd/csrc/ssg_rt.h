/*
 * ssg_rt.h -- thin runtime layer under the host orchestration: HIP (product) or the host wave
 * emulator (tests/emu, CPU-side tests only).  Device memory, copies, launches.
 */
#ifndef SSG_RT_H
#define SSG_RT_H
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <string>

extern thread_local std::string ssg_err_msg;

#ifdef SSG_EMU
#include "emu.h"
#define SSG_BACKEND "emu"
/* SSG_EMU_DEVICES = n pretends n devices (multi-device host logic under test); they share the host's memory */
extern thread_local int ssg_cur_dev;
static inline int rt_device_count() { const char *e = getenv("SSG_EMU_DEVICES"); const int n = e ? atoi(e) : 1; return n > 0 ? n : 1; }
static inline int rt_set_device(int d) { if (d < 0 || d >= rt_device_count()) { ssg_err_msg = "ssg_set_device: no such device"; return -22; } ssg_cur_dev = d; return 0; }
#define SSG_MAX_LANE 8
extern thread_local int ssg_lane;
static inline int rt_set_lane(int l) { if (l < 0 || l >= SSG_MAX_LANE) { ssg_err_msg = "ssg_set_lane: lane out of range"; return -22; } ssg_lane = l; return 0; }
/* SSG_EMU_POISON=1: device memory comes filled with 0xA5 instead of zeros (the device pool hands out whatever the last user left): reads of
 * memory nobody wrote show in the CPU-side suite */
static inline void *rt_malloc(size_t n) { static const int poison = getenv("SSG_EMU_POISON") ? atoi(getenv("SSG_EMU_POISON")) : 0; if (!poison) return calloc(n ? n : 1, 1); void *p = malloc(n ? n : 1); if (p) memset(p, 0xA5, n ? n : 1); return p; }
static inline void rt_free(void *p) { free(p); }
static inline int rt_h2d(void *d, const void *h, size_t n) { if (n) memcpy(d, h, n); return 0; }
static inline int rt_d2h(void *h, const void *d, size_t n) { if (n) memcpy(h, d, n); return 0; }
static inline int rt_memset(void *d, int v, size_t n) { if (n) memset(d, v, n); return 0; }
static inline int rt_sync() { return 0; }
static inline void *rt_malloc_raw(size_t n) { return calloc(n ? n : 1, 1); }
static inline void rt_free_raw(void *p) { free(p); }
static inline void rt_pool_release() {}
static inline int rt_d2d(void *d, const void *s, size_t n) { if (n) memmove(d, s, n); return 0; }
static inline void *rt_host_alloc(size_t n) { return malloc(n ? n : 1); }
static inline void rt_host_free(void *p) { free(p); }
#define SSG_LAUNCH(kern, grid, block, lds, ...) do { if ((grid) > 0) emu::launch((unsigned)(grid), (unsigned)(block), (lds), [&]() { kern(__VA_ARGS__); }); } while (0)
#define SSG_LAUNCH_ON(si, kern, grid, block, lds, ...) SSG_LAUNCH(kern, grid, block, lds, __VA_ARGS__)
static inline void ssg_fork(int) {}
static inline void ssg_join(int) {}
static inline void ssg_prof_flush() {}
#else
#include <hip/hip_runtime.h>
#define SSG_BACKEND "hip:gfx950"
static inline int rt_check(hipError_t e, const char *what)
{
	if (e == hipSuccess) return 0;
	ssg_err_msg = std::string(what) + ": " + hipGetErrorString(e);
	return -1000;
}
static inline int rt_device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }
/* one host thread drives one device at a time (hipSetDevice is per thread); every piece of per-device state below -- HBM arena, side
 * streams -- is selected by the calling thread's current device, so N threads can drive N devices through the same entry points */
#define SSG_MAX_DEV 16
extern thread_local int ssg_cur_dev;
/* Lanes: several calls in flight on ONE device (ssg_set_lane, bin/bwa's SSG_BWA_INFLIGHT).  Lane 0 is the default stream, one call at a
 * time per device, as ever.  A thread on lane k > 0 does everything -- launches, copies, fills, library primitives, the fork / join of
 * its side streams -- on the stream of (device, k), waits only for that stream, and allocates from an arena of its own: an arena
 * hands a freed block to the next request in stream order, which holds within a lane and not across two. */
#define SSG_MAX_LANE 8
extern thread_local int ssg_lane;
extern thread_local hipStream_t ssg_stream;
#include <mutex>
static inline hipStream_t ssg_lane_stream(int dev, int lane)
{
	static hipStream_t s[SSG_MAX_DEV][SSG_MAX_LANE]; static std::mutex mu;
	if (lane <= 0) return 0;
	std::lock_guard<std::mutex> l(mu);
	if (!s[dev][lane]) (void)hipStreamCreateWithFlags(&s[dev][lane], hipStreamNonBlocking);
	return s[dev][lane];
}
static inline int rt_set_device(int d) { if (d < 0 || d >= SSG_MAX_DEV) { ssg_err_msg = "ssg_set_device: device index out of range"; return -22; } int rc = rt_check(hipSetDevice(d), "hipSetDevice"); if (!rc) { ssg_cur_dev = d; ssg_stream = ssg_lane_stream(d, ssg_lane); } return rc; }
static inline int rt_set_lane(int l) { if (l < 0 || l >= SSG_MAX_LANE) { ssg_err_msg = "ssg_set_lane: lane out of range"; return -22; } ssg_lane = l; ssg_stream = ssg_lane_stream(ssg_cur_dev, l); if (l && !ssg_stream) { ssg_err_msg = "ssg_set_lane: cannot create a stream"; return -1000; } return 0; }
/* HBM arena: freed blocks are kept in size-class free lists and reused by later calls, so the
 * steady-state hot path performs no hipMalloc/hipFree (288 GB of HBM3E make the slack irrelevant) */
#include <algorithm>
#include <map>
#include <unordered_map>
#include <mutex>
struct ssg_pool_t {
	std::mutex mu; std::map<size_t, std::vector<void*> > free_; std::unordered_map<void*, size_t> size_;
	size_t held = 0, free_bytes = 0, held_max = 0;   /* bytes this arena got from the driver and has not given back; of them in the free lists */
	/* size classes: powers of two up to 64 MB; above, eight steps per octave (a request is rounded up by at most an eighth).  The per-batch
	 * arrays of a whole-genome run differ by a few per cent from call to call: with finer classes (64 MB steps until round 5) every array
	 * left a block in several neighbouring classes and the free lists grew until another process of the pipeline could not allocate. */
	static size_t cls(size_t n) { size_t c = 256; while (c < n) c <<= 1; if (c > (64u << 20)) { const size_t g = c >> 4; c = (n + g - 1) / g * g; } return c; }
	/* what the free lists may hold (SSG_POOL_FREE_GB, default 48: about what the arrays of one device call of 1 M pairs add up to); beyond it a
	 * returned block goes back to the driver, largest classes first */
	static size_t free_cap() { static const size_t c = (size_t)(std::max(getenv("SSG_POOL_FREE_GB") && atof(getenv("SSG_POOL_FREE_GB")) > 0 ? atof(getenv("SSG_POOL_FREE_GB")) : 48.0, 0.25) * 1073741824.0); return c; }   /* (fractions of a GB count; never below 256 MB: a cap of 0 would send every free to hipFree, which waits for the device) */
	std::unordered_map<void*, int> live_;   /* SSG_POOL_CHECK=1: a block handed out twice, or given back twice, is reported */
	static bool check() { static const int c = getenv("SSG_POOL_CHECK") ? atoi(getenv("SSG_POOL_CHECK")) : 0; return c != 0; }
	~ssg_pool_t() { if (held_max && getenv("SSG_POOL_LOG")) fprintf(stderr, "[ssgpu] device arena: at most %.2f GB held, %.2f GB of it in the free lists at exit\n", held_max / 1073741824.0, free_bytes / 1073741824.0); }
	void *get(size_t n) {
		size_t c = cls(n ? n : 1);
		{ std::lock_guard<std::mutex> l(mu); auto it = free_.find(c); if (it != free_.end() && !it->second.empty()) { void *p = it->second.back(); it->second.pop_back(); free_bytes -= c;
			if (check() && live_[p]++) fprintf(stderr, "[ssgpu] pool: block %p (class %zu) handed out while in use\n", p, c);
			return p; } }
		void *p = 0;
		if (hipMalloc(&p, c) != hipSuccess) { (void)hipGetLastError(); release(); if (hipMalloc(&p, c) != hipSuccess) { (void)hipGetLastError(); return 0; } }
		std::lock_guard<std::mutex> l(mu); size_[p] = c; held += c; held_max = held > held_max ? held : held_max; if (check()) live_[p] = 1; return p;
	}
	bool put(void *p) { std::lock_guard<std::mutex> l(mu); auto it = size_.find(p); if (it == size_.end()) return false;
		if (check() && --live_[p] != 0) fprintf(stderr, "[ssgpu] pool: block %p (class %zu) given back twice\n", p, it->second);
		free_[it->second].push_back(p); free_bytes += it->second;
		while (free_bytes > free_cap() && !free_.empty()) {   /* trim: the largest free block first */
			auto big = free_.end(); --big;
			if (big->second.empty()) { free_.erase(big); continue; }
			void *q = big->second.back(); big->second.pop_back(); free_bytes -= big->first; held -= big->first; size_.erase(q); (void)hipFree(q);
		}
		return true; }
	void release() { std::lock_guard<std::mutex> l(mu); for (auto &kv : free_) for (void *p : kv.second) { size_.erase(p); held -= kv.first; (void)hipFree(p); } free_.clear(); free_bytes = 0; }
};
extern ssg_pool_t ssg_pools[SSG_MAX_DEV][SSG_MAX_LANE];
#define ssg_pool (ssg_pools[ssg_cur_dev][ssg_lane])
static inline void *rt_malloc(size_t n) { return ssg_pool.get(n); }
static inline void rt_free(void *p)
{	/* back to the arena of the device it came from (normally the caller's) */
	if (!p) return;
	if (ssg_pool.put(p)) return;
	for (int d = 0; d < SSG_MAX_DEV; ++d) for (int l = 0; l < SSG_MAX_LANE; ++l) if ((d != ssg_cur_dev || l != ssg_lane) && ssg_pools[d][l].put(p)) return;
	(void)hipFree(p);
}
static inline int rt_lane_copy(void *d, const void *s, size_t n, hipMemcpyKind k, const char *what)
{	/* on a lane: queued on its stream, and the host side may touch its buffer again when this returns */
	int rc = rt_check(hipMemcpyAsync(d, s, n, k, ssg_stream), what);
	return rc ? rc : rt_check(hipStreamSynchronize(ssg_stream), what);
}
static inline int rt_h2d(void *d, const void *h, size_t n) { return !n ? 0 : ssg_stream ? rt_lane_copy(d, h, n, hipMemcpyHostToDevice, "hipMemcpyAsync H2D") : rt_check(hipMemcpy(d, h, n, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
static inline int rt_d2h(void *h, const void *d, size_t n) { return !n ? 0 : ssg_stream ? rt_lane_copy(h, d, n, hipMemcpyDeviceToHost, "hipMemcpyAsync D2H") : rt_check(hipMemcpy(h, d, n, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
static inline int rt_memset(void *d, int v, size_t n) { return !n ? 0 : ssg_stream ? rt_check(hipMemsetAsync(d, v, n, ssg_stream), "hipMemsetAsync") : rt_check(hipMemset(d, v, n), "hipMemset"); }
/* waits for THIS thread's stream (lane 0: the null stream), not for the device: kernels forked onto the side streams (which are joined by events before their results are
 * used) keep running across the host round trips of the stream that forked them -- a device-wide wait here made the chaining stage's rank preparation wait for the light
 * reads' kernels (6 ms of a 20 ms stage, profiles/r05q_chain_stage_timeline_before.txt) */
static inline int rt_sync() { int rc = rt_check(hipStreamSynchronize(ssg_stream), "hipStreamSynchronize"); return rc ? rc : rt_check(hipGetLastError(), "kernel launch"); }
/* multi-gigabyte, build-time-only arrays (index construction) bypass the arena: they must return to the driver when freed */
static inline void *rt_malloc_raw(size_t n) { void *p = 0; if (hipMalloc(&p, n ? n : 1) != hipSuccess) { (void)hipGetLastError(); ssg_pool.release(); if (hipMalloc(&p, n ? n : 1) != hipSuccess) { (void)hipGetLastError(); return 0; } } return p; }
static inline void rt_free_raw(void *p) { if (p) (void)hipFree(p); }
static inline void rt_pool_release() { ssg_pool.release(); }
static inline int rt_d2d(void *d, const void *s, size_t n) { return !n ? 0 : ssg_stream ? rt_check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, ssg_stream), "hipMemcpyAsync D2D") : rt_check(hipMemcpy(d, s, n, hipMemcpyDeviceToDevice), "hipMemcpy D2D"); }
/* page-locked host memory for the results a call hands to its caller.  A pageable destination costs a staging copy plus first-touch
 * faults and zero-fill of a fresh multi-hundred-MB block per call; page-locked blocks copy at PCIe speed but are expensive to make,
 * so freed blocks wait here for the next call (at most 4 GB of them). */
struct ssg_hostpool_t {
	std::mutex mu; std::unordered_map<void*, size_t> cap_; std::vector<void*> free_; size_t free_bytes = 0;
	void *get(size_t n)
	{
		if (!n) n = 1;
		{	std::lock_guard<std::mutex> l(mu);
			int best = -1;
			for (int i = 0; i < (int)free_.size(); ++i) { const size_t c = cap_[free_[i]]; if (c >= n && c <= 2 * n + ((size_t)1 << 20) && (best < 0 || c < cap_[free_[best]])) best = i; }
			if (best >= 0) { void *p = free_[best]; free_.erase(free_.begin() + best); free_bytes -= cap_[p]; return p; }
		}
		void *p = 0; const size_t c = n + n / 4;
		if (hipHostMalloc(&p, c, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return 0; }
		if (getenv("SSG_HOSTPOOL_LOG")) fprintf(stderr, "[ssgpu] page-locked block of %.0f MB made (pool holds %.0f MB free)\n", (double)c / 1048576.0, (double)free_bytes / 1048576.0);
		std::lock_guard<std::mutex> l(mu); cap_[p] = c;
		return p;
	}
	void put(void *p)
	{
		if (!p) return;
		std::vector<void*> drop;
		{	std::lock_guard<std::mutex> l(mu);
			free_.push_back(p); free_bytes += cap_[p];
			static const size_t cap_bytes = []() { const char *e = getenv("SSG_HOSTPOOL_GB"); const double g = e && atof(e) > 0 ? atof(e) : 4.0; return (size_t)(g * 1073741824.0); }();
			while (free_bytes > cap_bytes && !free_.empty()) { void *q = free_.front(); free_.erase(free_.begin()); free_bytes -= cap_[q]; cap_.erase(q); drop.push_back(q); }
		}
		for (void *q : drop) (void)hipHostFree(q);
	}
};
extern ssg_hostpool_t ssg_hostpool;
static inline void *rt_host_alloc(size_t n) { return ssg_hostpool.get(n); }
static inline void rt_host_free(void *p) { ssg_hostpool.put(p); }
/* optional per-kernel timing: HIP events recorded on the launch stream (the default stream) */
#include <vector>
struct ssg_prof_rec { const char *name; hipEvent_t a, b; };
extern int ssg_prof_on;
extern thread_local std::vector<ssg_prof_rec> ssg_prof_pending;   /* launches of this host thread; ssg_prof_flush() moves them to the process-wide table */
void ssg_prof_flush();
#define SSG_LAUNCH(kern, grid, block, lds, ...) do { if ((grid) > 0) { \
	if (ssg_prof_on) { ssg_prof_rec r_; r_.name = #kern; (void)hipEventCreate(&r_.a); (void)hipEventCreate(&r_.b); (void)hipEventRecord(r_.a, ssg_stream); \
		hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (lds), ssg_stream, __VA_ARGS__); (void)hipEventRecord(r_.b, ssg_stream); ssg_prof_pending.push_back(r_); } \
	else hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (lds), ssg_stream, __VA_ARGS__); } } while (0)
/* side streams for independent kernels that each leave most of the chip idle (few heavy work items): fork after the work
 * already queued on the default stream, launch with SSG_LAUNCH_ON(i, ...), join before anything that consumes the results */
static inline hipStream_t ssg_side_stream(int i) { static hipStream_t s[SSG_MAX_DEV][SSG_MAX_LANE][8]; hipStream_t &x = s[ssg_cur_dev][ssg_lane][i]; if (!x) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking); return x; }   /* a (device, lane) belongs to one thread at a time */
static inline void ssg_fork(int n) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); (void)hipEventRecord(e, ssg_stream); for (int i = 0; i < n; ++i) (void)hipStreamWaitEvent(ssg_side_stream(i), e, 0); (void)hipEventDestroy(e); }
static inline void ssg_join(int n) { for (int i = 0; i < n; ++i) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); (void)hipEventRecord(e, ssg_side_stream(i)); (void)hipStreamWaitEvent(ssg_stream, e, 0); (void)hipEventDestroy(e); } }
#define SSG_LAUNCH_ON(si, kern, grid, block, lds, ...) do { if ((grid) > 0) { hipStream_t st_ = ssg_side_stream(si); \
	if (ssg_prof_on) { ssg_prof_rec r_; r_.name = #kern; (void)hipEventCreate(&r_.a); (void)hipEventCreate(&r_.b); (void)hipEventRecord(r_.a, st_); \
		hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (lds), st_, __VA_ARGS__); (void)hipEventRecord(r_.b, st_); ssg_prof_pending.push_back(r_); } \
	else hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (lds), st_, __VA_ARGS__); } } while (0)
#endif

/* RAII device buffer */
template <class T> struct dbuf {
	T *p; size_t n;
	dbuf() : p(0), n(0) {}
	explicit dbuf(size_t n_) : p((T*)rt_malloc(n_ * sizeof(T))), n(n_) {}
	~dbuf() { rt_free(p); }
	dbuf(const dbuf&) = delete; dbuf &operator=(const dbuf&) = delete;
	bool alloc(size_t n_) { rt_free(p); n = n_; p = (T*)rt_malloc(n_ * sizeof(T)); return p != 0; }
	bool ok() const { return p != 0; }
	void swap(dbuf &o) { T *tp = p; p = o.p; o.p = tp; size_t tn = n; n = o.n; o.n = tn; }
	int up(const T *h, size_t cnt) { return rt_h2d(p, h, cnt * sizeof(T)); }
	int down(T *h, size_t cnt) const { return rt_d2h(h, p, cnt * sizeof(T)); }
	int zero() { return rt_memset(p, 0, n * sizeof(T)); }
};
/* host-side result array in page-locked pooled memory (contents undefined after resize) */
template <class T> struct hbuf {
	T *p; size_t n, cap;
	hbuf() : p(0), n(0), cap(0) {}
	~hbuf() { rt_host_free(p); }
	hbuf(const hbuf&) = delete; hbuf &operator=(const hbuf&) = delete;
	bool resize(size_t n_) { if (n_ > cap) { rt_host_free(p); p = (T*)rt_host_alloc(n_ * sizeof(T)); cap = p ? n_ : 0; } n = p || !n_ ? n_ : 0; return p != 0 || n_ == 0; }
	T *data() { return p; }
	const T *data() const { return p; }
	size_t size() const { return n; }
};
#endif
