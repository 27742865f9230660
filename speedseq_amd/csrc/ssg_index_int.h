/*
 * ssg_index_int.h -- the FM-index object shared by the translation units of libssgpu
 * (ssgpu_core.cpp: load / use; ssg_index_build.cpp: construction, upstream `bwa index`).
 */
#ifndef SSG_INDEX_INT_H
#define SSG_INDEX_INT_H
#include <vector>
#include <string>
#include <mutex>
#include <string.h>
#include "ssg_types.h"

struct ssg_hole_t { int64_t offset; int32_t len; char amb; };   /* upstream bntamb1_t (.amb) */

struct ssg_index {
	ssg_index_view_t v;
	/* owned device arrays (NULL when borrowed from the caller, ssg_index_from_device) */
	uint32_t *bwt; uint64_t *sa; uint8_t *pac; int64_t *ctg_off; int32_t *ctg_len;
	/* intervals of every pattern of 1..ktab_k bases (k_smem2.h), handed to the seeding kernel as arguments of its own -- the by-value view every
	 * kernel takes stays as small as it is */
	uint64_t *ktab; int ktab_k;
	uint64_t bwt_words;                     /* u32 words of the .bwt body (0 when borrowed) */
	bool raw_alloc;                         /* bwt/sa/pac came from rt_malloc_raw (index builder) */
	std::vector<std::string> names, annos;  /* .ann: name and FASTA comment ("" = upstream's "(null)") */
	std::vector<int32_t> n_ambs;            /* .ann: holes per contig */
	std::vector<ssg_hole_t> holes;          /* .amb */
	std::vector<int64_t> h_off; std::vector<int32_t> h_len;
	/* the contig names in HBM, for the kernel that writes SA / XA strings (k_bam.h); made by the first call that needs them (ssg_bam.cpp), dropped by ssg_index_set_names */
	std::mutex names_mu; char *d_names; int32_t *d_name_off;
	ssg_index() : bwt(0), sa(0), pac(0), ctg_off(0), ctg_len(0), ktab(0), ktab_k(0), bwt_words(0), raw_alloc(false), d_names(0), d_name_off(0) { memset(&v, 0, sizeof(v)); }
};
/* ssg_seed.cpp: the product's seeding kernels (k_smem2.h), the table of short-pattern intervals (every constructor of an index ends with
 * ssg_index_build_ktab) and the self-check of the denser suffix-array copy (SSG_SA_VERIFY) */
extern "C" int ssg_index_build_ktab(ssg_index *ix);
extern "C" int ssg_seed_smem2(const ssg_index *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *d_seq, const int64_t *d_off, int max_len, int cap,
                              ssg_intv_t *d_intv, int32_t *d_n, unsigned long long *n_extend, unsigned int max_ext, uint32_t *d_n_ext_read);
extern "C" int ssg_sa_verify(const ssg_index *ix, int new_intv, const uint64_t *d_sa_new, long n_new);

/* Layout fingerprint of the declarations the translation units of libssgpu share (and that kernels take by value).  Every unit defines
 * one with SSG_ABI_FP_DEFINE(<unit>); ssg_abi_selfcheck() (ssgpu_core.cpp, run before the first index is made) compares them and refuses
 * to go on when two units were compiled against different declarations -- an object left over from before a header changed would
 * otherwise read another unit's structures at the wrong offsets without any diagnostic. */
#include <stddef.h>
struct ssg_abi_fp_t { uint32_t v[20]; };
#ifdef __clang__
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winvalid-offsetof"
#else
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
#endif
static inline ssg_abi_fp_t ssg_abi_fp_make()
{
	ssg_abi_fp_t f = {{ (uint32_t)sizeof(ssg_index_view_t), (uint32_t)sizeof(ssg_mem_opt_t), (uint32_t)sizeof(ssg_index), (uint32_t)sizeof(ssg_intv_t),
		(uint32_t)offsetof(ssg_index_view_t, primary), (uint32_t)offsetof(ssg_index_view_t, L2), (uint32_t)offsetof(ssg_index_view_t, l_pac), (uint32_t)offsetof(ssg_index_view_t, sa_intv),
		(uint32_t)offsetof(ssg_mem_opt_t, min_seed_len), (uint32_t)offsetof(ssg_mem_opt_t, split_width), (uint32_t)offsetof(ssg_mem_opt_t, max_mem_intv), (uint32_t)offsetof(ssg_mem_opt_t, split_factor),
		(uint32_t)offsetof(ssg_mem_opt_t, mat), (uint32_t)offsetof(ssg_index, bwt), (uint32_t)offsetof(ssg_index, ktab), (uint32_t)offsetof(ssg_index, bwt_words),
		(uint32_t)offsetof(ssg_index, names), (uint32_t)sizeof(ssg_seed_t), (uint32_t)sizeof(ssg_alnreg_t), (uint32_t)sizeof(ssg_aln_t) }};
	return f;
}
#ifdef __clang__
#pragma clang diagnostic pop
#else
#pragma GCC diagnostic pop
#endif
#define SSG_ABI_FP_DEFINE(unit) extern "C" void ssg_abi_fp_##unit(ssg_abi_fp_t *out) { *out = ssg_abi_fp_make(); }
extern "C" int ssg_abi_selfcheck(void);   /* 0, or SSG_EINVAL with the differing unit / field in ssg_last_error() */
#endif
