/*
 * ssg_seed.cpp -- the seeding stage's kernels in a translation unit of their own (bwt_extend, bwt_smem1a,
 * bwt_seed_strategy1, mem_collect_intv).  A small unit compiles in seconds, so variants of the kernel can be built side by side and
 * checked on the MI355X in one GPU call (`make variant VUNITS=ssg_seed`, tools/dbg/seed_variants.sh).
 */
#include <algorithm>
#include "ssg_rt.h"
#include "k_smem2.h"
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"

SSG_ABI_FP_DEFINE(seed)
#define CHK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
/* rows (interval lists of one backward step) longer than this send a read to the wave-per-read kernel; only with an extension budget */
static int budget_row(unsigned int max_ext) { return max_ext < 0x7fffffffu ? env_int("SSG_SMEM_MAX_ROW", 32) : 0x7fffffff; }

/* the instances of the lane kernel under names of their own (the launch macro takes one token and names the kernel in the per-kernel profile) */
static constexpr auto ssg_k_smem2_plain = ssg_k_smem2<false, false>;   /* no table of short-pattern intervals (SSG_KTAB_K=0, or an index too small for one) */
static constexpr auto ssg_k_smem2_kt = ssg_k_smem2<false, true>;       /* the product instance */
static constexpr auto ssg_k_smem2_tune = ssg_k_smem2<true, false>;     /* launch statistics (SSG_S2_TUNE=1) */
static constexpr auto ssg_k_smem2_kt_tune = ssg_k_smem2<true, true>;

/* ssg_k_smem2 over all reads: d_n[r] = interval count, -1 (a capacity overflowed) or -2 (given up at max_ext extensions); lists unsorted.
 * SSG_S2_TUNE=1 runs the instrumented instance and prints its launch statistics (tools/dbg/smem_timeline.py reads them). */
extern "C" int ssg_seed_smem2(const ssg_index *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *d_seq, const int64_t *d_off, int max_len, int cap,
                              ssg_intv_t *d_intv, int32_t *d_n, unsigned long long *n_extend, unsigned int max_ext, uint32_t *d_n_ext_read)
{
	const int block = 64;
	const long nthreads = std::min<long>(((long)n_reads + block - 1) / block * block, 256L * env_int("SSG_SMEM_WAVES_PER_CU", 16) * 64)   /* resident waves per CU, measured (round 4, 1 M pairs): with the table 52.2 ms at 16, 57.4 at 12, 72.6 at 8; without it 12 was best (58.0 vs 61.6 at 16) */;
	const int scap = max_len + 2;
	/* the table of short-pattern intervals: used when the index has one and its patterns are shorter than a seed (the third pass jumps kt_k bases in) */
	const int kt_want = idx->ktab && env_int("SSG_SMEM_USE_KTAB", 1) ? std::min(idx->ktab_k, opt->min_seed_len - 1) : 0;   /* (every level below the table's K is there too: `-k' at or below K uses the levels under it) */
	const int kt_k = kt_want >= 2 ? kt_want : 0;
	const ssg_pk2_t *const kt = kt_k ? (const ssg_pk2_t*)idx->ktab : (const ssg_pk2_t*)0;
	dbuf<ssg_pk2_t> scratch((size_t)nthreads * 2 * scap + 64);
	dbuf<unsigned int> d_next(4);   /* [0] next read of the lane kernel, [1] reads it gave up, [2] next of those for the wave kernel */
	const int max_row = budget_row(max_ext);
	const bool budget = max_ext < 0x7fffffffu;
	dbuf<int32_t> d_heavy(budget ? (size_t)n_reads : 1);
	if (!scratch.ok() || !d_next.ok() || !d_heavy.ok()) { ssg_err_msg = "device allocation failed: seeding work lists"; return SSG_ENOMEM; }
	CHK(d_next.zero());
	int32_t *const hv = budget ? d_heavy.p : (int32_t*)0;
	if (env_int("SSG_S2_TUNE", 0)) {
		unsigned long long st[64]; memset(st, 0, sizeof(st)); st[0] = st[1] = ~0ull;
#ifdef SSG_EMU
		memcpy(ssg_s2_stat, st, sizeof(st));
#else
		CHK(rt_sync()); CHK(rt_check(hipMemcpyToSymbol(HIP_SYMBOL(ssg_s2_stat), st, sizeof(st)), "hipMemcpyToSymbol"));
#endif
		if (kt) SSG_LAUNCH(ssg_k_smem2_kt_tune, nthreads / block, block, 0, idx->v, *opt, n_reads, (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, scratch.p, scap, n_extend, d_next.p, max_ext, max_row, d_n_ext_read, hv, d_next.p + 1, kt, kt_k);
		else SSG_LAUNCH(ssg_k_smem2_tune, nthreads / block, block, 0, idx->v, *opt, n_reads, (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, scratch.p, scap, n_extend, d_next.p, max_ext, max_row, d_n_ext_read, hv, d_next.p + 1, kt, kt_k);
		CHK(rt_sync());
#ifdef SSG_EMU
		memcpy(st, ssg_s2_stat, sizeof(st));
#else
		CHK(rt_check(hipMemcpyFromSymbol(st, HIP_SYMBOL(ssg_s2_stat), sizeof(st)), "hipMemcpyFromSymbol"));
#endif
		fprintf(stderr, "[ssgpu] smem2 timeline (ms): pool of reads empty at %.2f, last lane done at %.2f; %llu reads given up at %u extensions / rows of %d; rounds %llu, lanes ready %.1f, with a read %.1f of 64\n",
		        (double)(st[1] - st[0]) / 1e5, (double)(st[2] - st[0]) / 1e5, st[3], max_ext, max_row, st[48], st[48] ? (double)st[49] / st[48] : 0., st[48] ? (double)st[50] / st[48] : 0.);
		fprintf(stderr, "[ssgpu] smem2 reads by extensions (< 2^b):");
		for (int b = 1; b <= 24; ++b) if (st[8 + b]) fprintf(stderr, " b%d=%llu", b, st[8 + b]);
		fprintf(stderr, "\n");
	} else if (kt) SSG_LAUNCH(ssg_k_smem2_kt, nthreads / block, block, 0, idx->v, *opt, n_reads, (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, scratch.p, scap, n_extend, d_next.p, max_ext, max_row, d_n_ext_read, hv, d_next.p + 1, kt, kt_k);
	else SSG_LAUNCH(ssg_k_smem2_plain, nthreads / block, block, 0, idx->v, *opt, n_reads, (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, scratch.p, scap, n_extend, d_next.p, max_ext, max_row, d_n_ext_read, hv, d_next.p + 1, kt, kt_k);
	if (budget) {   /* the given-up reads, a wave each (their number stays on the device: the launch is sized for the chip) */
		const long n_wg = std::min<long>(n_reads, 256L * env_int("SSG_SMEM_HEAVY_WAVES_PER_CU", 28));
		if (scap <= 160) SSG_LAUNCH(ssg_k_smem_heavy<160>, n_wg, block, 0, idx->v, *opt, (const int32_t*)d_heavy.p, (const unsigned int*)(d_next.p + 1), (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, n_extend, d_next.p + 2);
		else if (scap <= 264) SSG_LAUNCH(ssg_k_smem_heavy<264>, n_wg, block, 0, idx->v, *opt, (const int32_t*)d_heavy.p, (const unsigned int*)(d_next.p + 1), (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, n_extend, d_next.p + 2);
		else SSG_LAUNCH(ssg_k_smem_heavy<328>, n_wg, block, 0, idx->v, *opt, (const int32_t*)d_heavy.p, (const unsigned int*)(d_next.p + 1), (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, n_extend, d_next.p + 2);
	}
	return rt_sync();
}

/* table of the intervals of all patterns up to K bases (k_smem2.h): K = SSG_KTAB_K, default 13 capped at log4(text length) - 2 (1.4 GB for a human-size index: 89 M
 * bwt_extend calls, ~20 ms); never more than 15 (s2_code packs 15 bases; 22.9 GB, < 50 ms) nor than log4(text length) + 2; 0 = none.  Every constructor of an index
 * ends with this.  Why not 15: with the device to itself the lane kernel takes 57.4 ms at 12, 56.0 at 13, 55.2 at 14, 52.6 at 15 (1 M pairs / 3.1 Gbp,
 * profiles/r06T_ktab_pairwave_sa_ab.json) -- but inside the script's pipeline, where two calls of `bwa mem' and two more processes share the memory system, the same
 * kernel takes 62 ms a launch at 13 and 100 at 15, and 40 M pairs go through at 1.57 against 1.42 M pairs/s (profiles/r06j2_soak_40M_ktab13.json, r06i2_..._ktab15.json,
 * one box): the small table's upper levels stay in the 256-MB cache, the large one's entries never do. */
extern "C" int ssg_index_build_ktab(ssg_index *ix)
{
	int lg = 0; while ((ix->v.seq_len >> (2 * (lg + 1))) != 0) ++lg;    /* floor(log4(seq_len)) */
	int K = env_int("SSG_KTAB_K", std::min(13, lg - 2));
	K = std::min(K, std::min(15, lg + 2));
	rt_free(ix->ktab); ix->ktab = 0; ix->ktab_k = 0;
	if (K < 2 || ix->v.seq_len >= (1ull << 40)) return 0;
	const size_t n_ent = (size_t)((((1ull << (2 * (K + 1))) - 4ull) / 3ull));
	ix->ktab = (uint64_t*)rt_malloc(n_ent * 16);
	if (!ix->ktab) { ssg_err_msg = "index allocation failed: table of short-pattern intervals"; return SSG_ENOMEM; }
	for (int j = 1; j <= K; ++j) { const long np = 1L << (2 * (j - 1)); SSG_LAUNCH(ssg_k_ktab_level, (np + 255) / 256, 256, 0, ix->v, j, (ssg_pk2_t*)ix->ktab); }
	CHK(rt_sync());
	if (env_int("SSG_KTAB_VERIFY", 0)) {
		dbuf<unsigned long long> d_bad(16); unsigned long long bad[16];
		if (!d_bad.ok()) { ssg_err_msg = "device allocation failed"; return SSG_ENOMEM; }
		CHK(d_bad.zero());
		for (int j = 1; j <= K; ++j) { const long np = 1L << (2 * j), stride = np > (1L << 22) ? np >> 22 : 1, nt = (np + stride - 1) / stride; SSG_LAUNCH(ssg_k_ktab_verify, (nt + 255) / 256, 256, 0, ix->v, j, stride, (const ssg_pk2_t*)ix->ktab, d_bad.p); }
		CHK(rt_sync()); CHK(d_bad.down(bad, 16));
		fprintf(stderr, "[ssgpu] table of short-pattern intervals, K = %d; entries that differ from forward extension, per level:", K);
		for (int j = 1; j <= K; ++j) fprintf(stderr, " %llu", bad[j]);
		fprintf(stderr, "\n");
	}
	ix->ktab_k = K;
	return 0;
}

/* SSG_SA_VERIFY: every `stride`-th entry of the denser SA table against upstream's bwt_sa on the file's samples (view = the index before the swap) */
extern "C" int ssg_sa_verify(const ssg_index *ix, int new_intv, const uint64_t *d_sa_new, long n_new)
{
	const long stride = n_new > (1L << 24) ? n_new >> 24 : 1, nt = (n_new + stride - 1) / stride;
	dbuf<unsigned long long> d_bad(1); unsigned long long bad = 0;
	if (!d_bad.ok()) { ssg_err_msg = "device allocation failed"; return SSG_ENOMEM; }
	CHK(d_bad.zero());
	SSG_LAUNCH(ssg_k_sa_verify, (nt + 255) / 256, 256, 0, ix->v, new_intv, d_sa_new, n_new, stride, d_bad.p);
	CHK(rt_sync()); CHK(d_bad.down(&bad, 1));
	fprintf(stderr, "[ssgpu] SA samples every %d rows: %llu of %ld checked entries differ from bwt_sa on the file's samples\n", new_intv, bad, nt);
	return 0;
}
