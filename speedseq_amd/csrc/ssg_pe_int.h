/*
 * ssg_pe_int.h -- what the translation units of libssgpu share about a call of the paired-end hot path: the host result, the result kept in
 * HBM, and the entry points of ssgpu_core.cpp that ssg_bam.cpp (k_bam.h: FASTQ text in, BAM record bytes out) builds on.
 */
#ifndef SSG_PE_INT_H
#define SSG_PE_INT_H
#include <vector>
#include "ssg_rt.h"
#include "ssg_types.h"
#include "ssg_index_int.h"

struct ssg_pe_result {
	int n_reads, n_batches, se = 0;      /* se: the reads are single-end (ssg_mem_process_reads): n_reads units, no pairs */
	std::vector<int64_t> req_off;        /* n_reads + 1 */
	hbuf<ssg_alnreq_t> req;              /* page-locked, recycled across calls */
	hbuf<ssg_aln_t> alns;
	std::vector<ssg_pestat_t> pes;        /* n_batches * 4 */
	uint64_t stats[8];
};

/* device-resident output of the PE stage (kept in HBM for the duplicate-marking stage / the bench / the BAM record kernel) */
struct pe_dev_t {
	dbuf<ssg_alnreq_t> req; dbuf<ssg_aln_t> alns; dbuf<int64_t> req_off; int64_t n_req;
};

/* ssgpu_core.cpp */
int ssg_need_device();
/* the whole PE hot path on device-resident inputs; `keep` != NULL leaves the records in HBM instead of downloading them into `res` */
int ssg_pe_core(const ssg_index *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *d_seq, const int64_t *d_off, int max_len,
                const int32_t *d_pair_batch, int n_batches, int64_t id0, const ssg_pestat_t *pes0, ssg_pe_result *res, pe_dev_t *keep);
/* exclusive prefix sum of n int32 counts into n + 1 int64 offsets, on the device; *total = out[n] */
int ssg_dev_exclusive_scan(const int32_t *d_in, int64_t *d_out, long n, int64_t *total);
#define SSG_MAX_READ_LEN 310   /* 2x300 with room; the kernels' column classes end at 320 (k_sw.h NS = 5, ssg_k_ext_lane<320>, SSG_S2_QWORDS) */
#endif
