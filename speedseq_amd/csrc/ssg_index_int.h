/*
 * ssg_index_int.h -- the FM-index object shared by the translation units of libssgpu
 * (ssgpu_core.cpp: load / use; ssg_index_build.cpp: construction, upstream `bwa index`).
 */
#ifndef SSG_INDEX_INT_H
#define SSG_INDEX_INT_H
#include <vector>
#include <string>
#include "ssg_types.h"

struct ssg_hole_t { int64_t offset; int32_t len; char amb; };   /* upstream bntamb1_t (.amb) */

struct ssg_index {
	ssg_index_view_t v;
	/* owned device arrays (NULL when borrowed from the caller, ssg_index_from_device) */
	uint32_t *bwt; uint64_t *sa; uint8_t *pac; int64_t *ctg_off; int32_t *ctg_len;
	/* optional (SSG_KTAB_K): bidirectional intervals of every pattern of 1..ktab_k bases, 16 bytes each (x0, x1, x2 in 40 bits), level j (4^j
	 * entries, little-endian base-4 pattern code) after the levels below it; handed to the seeding kernel's table instance as its own
	 * arguments -- the by-value view every kernel takes stays as it was in rounds 1-2 */
	uint64_t *ktab; int ktab_k;
	uint64_t bwt_words;                     /* u32 words of the .bwt body (0 when borrowed) */
	bool raw_alloc;                         /* bwt/sa/pac came from rt_malloc_raw (index builder) */
	std::vector<std::string> names, annos;  /* .ann: name and FASTA comment ("" = upstream's "(null)") */
	std::vector<int32_t> n_ambs;            /* .ann: holes per contig */
	std::vector<ssg_hole_t> holes;          /* .amb */
	std::vector<int64_t> h_off; std::vector<int32_t> h_len;
	ssg_index() : bwt(0), sa(0), pac(0), ctg_off(0), ctg_len(0), ktab(0), ktab_k(0), bwt_words(0), raw_alloc(false) { memset(&v, 0, sizeof(v)); }
};
/* ssg_ktab.cpp: the optional table (every constructor of an index ends with ssg_index_build_ktab), its seeding-kernel instance, the SA self-check */
extern "C" int ssg_index_build_ktab(ssg_index *ix);
extern "C" int ssg_ktab_launch_smem(const ssg_index *idx, const ssg_mem_opt_t *opt, long n_wg, int block, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                                    ssg_intv_t *d_intv, int32_t *d_n, int cap, ssg_intv_t *scratch, int scap, unsigned long long *n_extend, unsigned int *next_read);
extern "C" int ssg_sa_verify(const ssg_index *ix, int new_intv, const uint64_t *d_sa_new, long n_new);
#endif
