/*
 * k_swjobs.h -- stage-level job kernels for the local (ksw_align2, row a10) and global
 * (ksw_global2 + backtrace, row a12) Smith-Waterman primitives: one wavefront per job.
 */
#ifndef SSG_K_SWJOBS_H
#define SSG_K_SWJOBS_H
#include "k_sw.h"

__global__ void __launch_bounds__(256) ssg_k_align2_jobs(ssg_mem_opt_t opt, int n_jobs, const ssg_sw_job_t *jobs, const uint8_t *qbuf, const uint8_t *tbuf,
                                  ssg_kswr_t *res, unsigned long long *bscratch, int bstride)
{
	long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (wid >= n_jobs) return;
	ssg_sw_job_t jb = jobs[wid];
	ssg_seqv_t q = { qbuf + jb.qoff, 1 }, t = { tbuf + jb.toff, 1 };
	ssg_kswr_t r = wv_align2(opt, jb.qlen, q, jb.tlen, t, jb.xtra, bscratch + wid * (long)bstride, 0);
	if (wv_lane() == 0) res[wid] = r;
}

__global__ void __launch_bounds__(256) ssg_k_global_jobs(ssg_mem_opt_t opt, int n_jobs, const ssg_glb_job_t *jobs, const uint8_t *qbuf, const uint8_t *tbuf,
                                  int32_t *score, int32_t *n_cigar, uint32_t *cigar, int cap, uint8_t *zscratch, long zstride)
{
	long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (wid >= n_jobs) return;
	ssg_glb_job_t jb = jobs[wid];
	ssg_seqv_t q = { qbuf + jb.qoff, 1 }, t = { tbuf + jb.toff, 1 };
	uint8_t *z = zscratch + wid * zstride;
	int sc = wv_global2_any(opt, jb.qlen, q, jb.tlen, t, jb.w, z, 0);
	ssg_wave_memsync();
	if (wv_lane() == 0) {
		score[wid] = sc;
		n_cigar[wid] = ssg_global_backtrace(z, jb.qlen, jb.tlen, jb.w, cigar + wid * (long)cap, cap);
	}
}
#endif
