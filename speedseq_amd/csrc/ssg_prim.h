/*
 * ssg_prim.h -- device-wide primitives used by the host orchestration: rocPRIM (AMD's own device library; no CUB-compatibility
 * layer) on the MI355X; plain loops in the host-emulation build (tests/emu, CPU-side tests only).  64-bit item counts.
 */
#ifndef SSG_PRIM_H
#define SSG_PRIM_H
#include "ssg_rt.h"
#include "ssg_dev.h"
#ifndef SSG_EMU
#include <rocprim/rocprim.hpp>
#else
#include <algorithm>
#include <vector>
#endif

#define PRIM_TRY(call, what) do { if ((call) != hipSuccess) { ssg_err_msg = what " failed"; (void)hipGetLastError(); return -1000; } } while (0)

/* stable radix sort of (u64 key, u64 value) on key bits [b0, b1) */
static inline int prim_sort_pairs_u64(const uint64_t *k_in, uint64_t *k_out, const uint64_t *v_in, uint64_t *v_out, int64_t n, int b0, int b1)
{
	if (n <= 0) return 0;
#ifdef SSG_EMU
	const uint64_t mask = (b1 >= 64 ? ~0ull : (1ull << b1) - 1) & ~((1ull << b0) - 1);
	std::vector<int64_t> ix((size_t)n);
	for (int64_t i = 0; i < n; ++i) ix[(size_t)i] = i;
	std::stable_sort(ix.begin(), ix.end(), [&](int64_t a, int64_t b) { return (k_in[a] & mask) < (k_in[b] & mask); });
	for (int64_t i = 0; i < n; ++i) { k_out[i] = k_in[ix[(size_t)i]]; v_out[i] = v_in[ix[(size_t)i]]; }
	return 0;
#else
	size_t tb = 0;
	PRIM_TRY(rocprim::radix_sort_pairs(nullptr, tb, k_in, k_out, v_in, v_out, (size_t)n, (unsigned)b0, (unsigned)b1), "rocprim radix_sort_pairs (size query)");
	void *tmp = rt_malloc(tb + 16);
	if (!tmp) { ssg_err_msg = "device allocation failed: sort temporaries"; return -12; }
	hipError_t e = rocprim::radix_sort_pairs(tmp, tb, k_in, k_out, v_in, v_out, (size_t)n, (unsigned)b0, (unsigned)b1, ssg_stream);
	int rc = rt_sync(); rt_free(tmp);
	if (e != hipSuccess) { ssg_err_msg = "rocprim radix_sort_pairs failed"; (void)hipGetLastError(); return -1000; }
	return rc;
#endif
}

#ifndef SSG_EMU
struct prim_max_i64 { SSG_DEVMEM int64_t operator()(const int64_t &a, const int64_t &b) const { return a > b ? a : b; } };
struct prim_u32_to_u64 { SSG_DEVMEM uint64_t operator()(const uint32_t &v) const { return (uint64_t)v; } };
struct prim_u8_to_u64 { SSG_DEVMEM uint64_t operator()(const uint8_t &v) const { return (uint64_t)v; } };
#endif

/* inclusive running maximum of int64 */
static inline int prim_scan_max_i64(const int64_t *in, int64_t *out, int64_t n)
{
	if (n <= 0) return 0;
#ifdef SSG_EMU
	int64_t m = in[0];
	for (int64_t i = 0; i < n; ++i) { m = in[i] > m ? in[i] : m; out[i] = m; }
	return 0;
#else
	size_t tb = 0;
	PRIM_TRY(rocprim::inclusive_scan(nullptr, tb, in, out, (size_t)n, prim_max_i64()), "rocprim inclusive_scan (size query)");
	void *tmp = rt_malloc(tb + 16);
	if (!tmp) { ssg_err_msg = "device allocation failed: scan temporaries"; return -12; }
	hipError_t e = rocprim::inclusive_scan(tmp, tb, in, out, (size_t)n, prim_max_i64(), ssg_stream);
	int rc = rt_sync(); rt_free(tmp);
	if (e != hipSuccess) { ssg_err_msg = "rocprim inclusive_scan failed"; (void)hipGetLastError(); return -1000; }
	return rc;
#endif
}

/* exclusive prefix sums of u32 counts into n + 1 u64 offsets (out[n] = total) */
static inline int prim_exsum_u32_u64(const uint32_t *in, uint64_t *out, int64_t n)
{
#ifdef SSG_EMU
	uint64_t t = 0;
	for (int64_t i = 0; i < n; ++i) { out[i] = t; t += in[i]; }
	out[n] = t;
	return 0;
#else
	/* scan n + 1 items of an input that is zero-extended by one element: the counts array is allocated with one spare slot */
	rocprim::transform_iterator<const uint32_t*, prim_u32_to_u64, uint64_t> it(in, prim_u32_to_u64());
	size_t tb = 0;
	PRIM_TRY(rocprim::exclusive_scan(nullptr, tb, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>()), "rocprim exclusive_scan (size query)");
	void *tmp = rt_malloc(tb + 16);
	if (!tmp) { ssg_err_msg = "device allocation failed: scan temporaries"; return -12; }
	hipError_t e = rocprim::exclusive_scan(tmp, tb, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), ssg_stream);
	int rc = rt_sync(); rt_free(tmp);
	if (e != hipSuccess) { ssg_err_msg = "rocprim exclusive_scan failed"; (void)hipGetLastError(); return -1000; }
	return rc;
#endif
}

/* number of non-zero flags */
static inline int prim_count_flags(const uint8_t *flag, int64_t n, uint64_t *count)
{
	*count = 0;
	if (n <= 0) return 0;
#ifdef SSG_EMU
	uint64_t c = 0; for (int64_t i = 0; i < n; ++i) c += flag[i] != 0;
	*count = c; return 0;
#else
	rocprim::transform_iterator<const uint8_t*, prim_u8_to_u64, uint64_t> it(flag, prim_u8_to_u64());
	uint64_t *d_out = (uint64_t*)rt_malloc(8);
	size_t tb = 0;
	PRIM_TRY(rocprim::reduce(nullptr, tb, it, d_out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>()), "rocprim reduce (size query)");
	void *tmp = rt_malloc(tb + 16);
	if (!tmp || !d_out) { ssg_err_msg = "device allocation failed: reduce temporaries"; return -12; }
	hipError_t e = rocprim::reduce(tmp, tb, it, d_out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), ssg_stream);
	int rc = rt_sync();
	if (!rc) rc = rt_d2h(count, d_out, 8);
	rt_free(tmp); rt_free(d_out);
	if (e != hipSuccess) { ssg_err_msg = "rocprim reduce failed"; (void)hipGetLastError(); return -1000; }
	return rc;
#endif
}

/* out[] = in[i] for every i with flag[i] != 0, in order; in == NULL selects the indices base + i themselves */
static inline int prim_select_u64(const uint64_t *in, uint64_t base, const uint8_t *flag, int64_t n, uint64_t *out, uint64_t *n_out)
{
	*n_out = 0;
	if (n <= 0) return 0;
#ifdef SSG_EMU
	uint64_t c = 0;
	for (int64_t i = 0; i < n; ++i) if (flag[i]) out[c++] = in ? in[i] : base + (uint64_t)i;
	*n_out = c; return 0;
#else
	uint64_t *d_cnt = (uint64_t*)rt_malloc(8);
	if (!d_cnt) { ssg_err_msg = "device allocation failed: select count"; return -12; }
	size_t tb = 0; hipError_t e;
	rocprim::counting_iterator<uint64_t> cit(base);
	if (in) { PRIM_TRY(rocprim::select(nullptr, tb, in, flag, out, d_cnt, (size_t)n), "rocprim select (size query)"); }
	else { PRIM_TRY(rocprim::select(nullptr, tb, cit, flag, out, d_cnt, (size_t)n), "rocprim select (size query)"); }
	void *tmp = rt_malloc(tb + 16);
	if (!tmp) { rt_free(d_cnt); ssg_err_msg = "device allocation failed: select temporaries"; return -12; }
	if (in) e = rocprim::select(tmp, tb, in, flag, out, d_cnt, (size_t)n, ssg_stream);
	else e = rocprim::select(tmp, tb, cit, flag, out, d_cnt, (size_t)n, ssg_stream);
	int rc = rt_sync();
	if (!rc) rc = rt_d2h(n_out, d_cnt, 8);
	rt_free(tmp); rt_free(d_cnt);
	if (e != hipSuccess) { ssg_err_msg = "rocprim select failed"; (void)hipGetLastError(); return -1000; }
	return rc;
#endif
}
#endif
