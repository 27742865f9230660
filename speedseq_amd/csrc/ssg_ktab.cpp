/*
 * ssg_ktab.cpp -- optional table of short-pattern FM-index intervals and the seeding kernel that uses it (SURVEY.md 8a rows a1-a2;
 * SSG_KTAB_K = K switches it on, default off), plus the device self-checks of the index load (SSG_KTAB_VERIFY, SSG_SA_VERIFY).
 * A translation unit of its own on purpose: see k_seed_kt.h.  Host entry points are declared in ssg_index_int.h.
 */
#include <algorithm>
#include "ssg_rt.h"
#include "k_seed_kt.h"
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"

SSG_ABI_FP_DEFINE(ktab)
#define CHK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }

/* table of the intervals of all patterns up to K bases (ssg_index.ktab): K = SSG_KTAB_K (1.4 GB for K = 13; built level by level with
 * upstream's bwt_extend, ~90 M extensions) */
extern "C" int ssg_index_build_ktab(ssg_index *ix)
{
	int K = env_int("SSG_KTAB_K", 0);
	if (K > 14) K = 14;
	rt_free(ix->ktab); ix->ktab = 0; ix->ktab_k = 0;
	if (K < 1 || ix->v.seq_len >= (1ull << 40)) return 0;
	const size_t n_ent = (size_t)((((1ull << (2 * (K + 1))) - 4ull) / 3ull));
	ix->ktab = (uint64_t*)rt_malloc(n_ent * 16);
	if (!ix->ktab) { ssg_err_msg = "index allocation failed: k-mer interval table"; return SSG_ENOMEM; }
	const bool fwd = getenv("SSG_KTAB_BUILD") && !strcmp(getenv("SSG_KTAB_BUILD"), "fwd");
	for (int j = 1; j <= K; ++j) {
		const long np = 1L << (2 * (j - 1));
		if (fwd) SSG_LAUNCH(ssg_k_ktab_level_fwd, (4 * np + 255) / 256, 256, 0, ix->v, j, (ssg_pk_t*)ix->ktab);
		else SSG_LAUNCH(ssg_k_ktab_level, (np + 255) / 256, 256, 0, ix->v, j, (ssg_pk_t*)ix->ktab);
	}
	CHK(rt_sync());
	if (env_int("SSG_KTAB_VERIFY", 0)) {
		dbuf<unsigned long long> d_bad(16); unsigned long long bad[16];
		if (!d_bad.ok()) { ssg_err_msg = "device allocation failed"; return SSG_ENOMEM; }
		CHK(d_bad.zero());
		for (int j = 1; j <= K; ++j) { const long np = 1L << (2 * j), stride = np > (1L << 22) ? np >> 22 : 1, nt = (np + stride - 1) / stride; SSG_LAUNCH(ssg_k_ktab_verify, (nt + 255) / 256, 256, 0, ix->v, j, stride, (const ssg_pk_t*)ix->ktab, d_bad.p); }
		CHK(rt_sync()); CHK(d_bad.down(bad, 16));
		fprintf(stderr, "[ssgpu] k-mer interval table K=%d, entries differing from forward extension per level:", K);
		for (int j = 1; j <= K; ++j) fprintf(stderr, " %llu", bad[j]);
		fprintf(stderr, "\n");
	}
	ix->ktab_k = K;
	return 0;
}

/* the seeding kernel's table instance (lane per read); arguments as ssg_k_smem_quad<1> */
extern "C" int ssg_ktab_launch_smem(const ssg_index *idx, const ssg_mem_opt_t *opt, long n_wg, int block, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                                    ssg_intv_t *d_intv, int32_t *d_n, int cap, ssg_intv_t *scratch, int scap, unsigned long long *n_extend, unsigned int *next_read)
{
	SSG_LAUNCH(ssg_k_smem_quad_kt<1>, n_wg, block, 0, idx->v, *opt, n_reads, (const int32_t*)0, d_seq, d_off, d_intv, d_n, cap, scratch, scap, n_extend, next_read,
	           (const ssg_pk_t*)idx->ktab, idx->ktab_k);
	return 0;
}

#ifdef SSG_KT_DBG
/* diagnostic builds only: the counters / kernarg echo of ssg_k_smem_quad_kt (64 words), cleared by the read */
extern "C" int ssg_ktab_dbg_read(unsigned long long *out)
{
#ifdef SSG_EMU
	memcpy(out, ssg_kt_dbg, sizeof(ssg_kt_dbg)); memset(ssg_kt_dbg, 0, sizeof(ssg_kt_dbg));
#else
	CHK(rt_sync());
	CHK(rt_check(hipMemcpyFromSymbol(out, HIP_SYMBOL(ssg_kt_dbg), 64 * sizeof(unsigned long long)), "hipMemcpyFromSymbol"));
	unsigned long long z[64]; memset(z, 0, sizeof(z));
	CHK(rt_check(hipMemcpyToSymbol(HIP_SYMBOL(ssg_kt_dbg), z, sizeof(z)), "hipMemcpyToSymbol"));
#endif
	return 0;
}
#endif

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
