/*
 * k_swjobs.h -- stage-level job kernels for the local (ksw_align2, row a10) and global
 * (ksw_global2 + backtrace, row a12) Smith-Waterman primitives: one wavefront per job.
 */
#ifndef SSG_K_SWJOBS_H
#define SSG_K_SWJOBS_H
#include "k_sw.h"
#include "k_extlane.h"
#include "k_mswlane.h"

__global__ void __launch_bounds__(256) ssg_k_align2_jobs(ssg_mem_opt_t opt, int n_jobs, const ssg_sw_job_t *jobs, const uint8_t *qbuf, const uint8_t *tbuf,
                                  ssg_kswr_t *res, unsigned long long *bscratch, int bstride)
{
	long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (wid >= n_jobs) return;
	ssg_sw_job_t jb = jobs[wid];
	ssg_seqv_t q = { qbuf + jb.qoff, 1 }, t = { tbuf + jb.toff, 1 };
	ssg_kswr_t r = wv_align2<true>(opt, jb.qlen, q, jb.tlen, t, jb.xtra, bscratch + wid * (long)bstride, 0);
	if (wv_lane() == 0) res[wid] = r;
}

/* Stage-level twin of ssg_k_ext_lane (row a7, the kernel that runs mem_chain2aln's ksw_extend2 calls in the product path): one LANE per
 * job through the same ln_extend2 -- DP row in LDS words (13-bit h / e, 6-bit score table), target bases from a 2-bit pac.  A job's
 * target is `tlen' bases of the doubled coordinate system of the pac in `ix' from position toff in direction `dir' (+1 / -1), as the
 * product's left / right extensions read them; the query is qbuf[qoff .. qoff + qlen). */
template <int QCAP>
__global__ void __launch_bounds__(64) ssg_k_ext_lane_jobs(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_jobs, const ssg_ext_job_t *jobs, const int64_t *tpos, int dir,
                                  const uint8_t *qbuf, ssg_ext_res_t *res, unsigned long long *cells)
{
	constexpr int U = QCAP > 72 ? 4 : 2;
	__shared__ uint32_t L[(QCAP + 2 * U) * 64];
	const long g = (long)blockIdx.x * 64 + threadIdx.x;
	if (g >= n_jobs) return;
	const ssg_ext_job_t jb = jobs[g];
	uint32_t *Lc = L + (threadIdx.x & 63);
	if (jb.qlen > QCAP) return;
	for (int j = 0; j < jb.qlen; ++j) Lc[j * 64] = SSG_XL_QWORD(qbuf[jb.qoff + j]);
	Lc[jb.qlen * 64] = 0;
	unsigned long long nc = 0;
	res[g] = ln_extend2<U>(opt, ix, Lc, jb.qlen, jb.tlen, tpos[g], dir, jb.w, jb.end_bonus, jb.zdrop, jb.h0, &nc);
	if (cells && nc) atomicAdd(cells, nc);
}

/* Stage-level twin of mate rescue's alignment path (row a10; k_mswlane.h + k_pair.h wv_matesw): the forward pass of job i comes from the
 * lane kernel's slot fwd[i] when it is there (state 1, same window), the reverse pass too when the lane kernel ran it (state 2), else that pass -- or the whole call -- from the wave code, on the
 * job's window decoded from the index's 2-bit reference: tlen bases from doubled coordinate tpos[i]. */
__global__ void __launch_bounds__(256) ssg_k_align2_fin_jobs(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_jobs, const ssg_sw_job_t *jobs, const int64_t *tpos, const uint8_t *qbuf,
                                  const ssg_msres_t *fwd, ssg_kswr_t *res, uint8_t *tglb, int tstride, unsigned long long *bscratch, int bstride, int32_t *from_lane)
{
	long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (wid >= n_jobs) return;
	const ssg_sw_job_t jb = jobs[wid];
	uint8_t *tb = tglb + wid * (long)tstride;
	const int64_t rb = tpos[wid];
	const int pre = wv_get(fwd[wid].state >= 1 && fwd[wid].rb == rb && fwd[wid].tlen == jb.tlen ? fwd[wid].state : 0, 0);
	ssg_kswr_t r;
	if (pre) { r.score = wv_get(fwd[wid].score, 0); r.te = wv_get(fwd[wid].te, 0); r.qe = wv_get(fwd[wid].qe, 0); r.score2 = wv_get(fwd[wid].score2, 0); r.te2 = wv_get(fwd[wid].te2, 0); r.tb = r.qb = -1; }
	if (pre == 2) { r.tb = wv_get(fwd[wid].tb, 0); r.qb = wv_get(fwd[wid].qb, 0); }
	const bool want_rev = !pre || (pre == 1 && ssg_align2_has_rev(jb.xtra, r.score));
	ssg_wave_memsync();
	if (want_rev) for (int k = wv_lane(); k < (pre ? r.te + 1 : jb.tlen); k += 64) tb[k] = (uint8_t)ssg_ref_base(ix, rb + k);
	ssg_wave_memsync();
	ssg_seqv_t q = { qbuf + jb.qoff, 1 }, t = { tb, 1 };
	if (!pre) r = wv_align2<true>(opt, jb.qlen, q, jb.tlen, t, jb.xtra, bscratch + wid * (long)bstride, 0);
	else if (want_rev) wv_align2_rev<true>(opt, jb.qlen, q, t, jb.xtra, r, bscratch + wid * (long)bstride, 0);
	if (wv_lane() == 0) { res[wid] = r; from_lane[wid] = pre; }
}

__global__ void __launch_bounds__(256) ssg_k_global_jobs(ssg_mem_opt_t opt, int n_jobs, const ssg_glb_job_t *jobs, const uint8_t *qbuf, const uint8_t *tbuf,
                                  int32_t *score, int32_t *n_cigar, uint32_t *cigar, int cap, uint8_t *zscratch, long zstride)
{
	long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (wid >= n_jobs) return;
	ssg_glb_job_t jb = jobs[wid];
	ssg_seqv_t q = { qbuf + jb.qoff, 1 }, t = { tbuf + jb.toff, 1 };
	uint8_t *z = zscratch + wid * zstride;
	int sc = wv_global2_any<true>(opt, jb.qlen, q, jb.tlen, t, jb.w, z, 0);
	ssg_wave_memsync();
	if (wv_lane() == 0) {
		score[wid] = sc;
		n_cigar[wid] = ssg_global_backtrace(z, jb.qlen, jb.tlen, jb.w, cigar + wid * (long)cap, cap);
	}
}
#endif
