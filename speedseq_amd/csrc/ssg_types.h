/*
 * ssg_types.h -- plain-old-data types shared by the HIP kernels, the host orchestration and the
 * C-ABI of libssgpu (MI355X-native `speedseq align` hot path).
 *
 * Naming follows the reference's domain (upstream bwa: bwtintv_t, mem_seed_t, mem_chain_t,
 * mem_alnreg_t, mem_pestat_t, mem_aln_t).
 */
#ifndef SSG_TYPES_H
#define SSG_TYPES_H
#include <stdint.h>

#define SSG_MAX_INS_HIST 10001   /* insert-size histogram bins: 0..max_ins (mem_pestat, a9) */

/* upstream mem_opt_t (bwamem.h) -- same field meaning, same defaults (ssg_mem_opt_init) */
typedef struct {
	int32_t a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, pen_clip5, pen_clip3, w, zdrop;
	int32_t T, min_seed_len, min_chain_weight, max_chain_extend;
	int32_t split_width, max_occ, max_chain_gap, max_ins, max_matesw, max_XA_hits, max_XA_hits_alt;
	int32_t mapQ_coef_fac, chunk_size, n_threads;
	uint64_t max_mem_intv;
	float split_factor, mask_level, drop_ratio, XA_drop_ratio, mask_level_redun, mapQ_coef_len;
	int8_t mat[25];
	int8_t _pad0;
	uint16_t flag;          /* upstream opt->flag, the bits this path reads: SSG_F_NO_MULTI (-M), SSG_F_SOFTCLIP (-Y) */
	int8_t _pad[4];
} ssg_mem_opt_t;
#define SSG_F_NO_MULTI 0x10    /* upstream MEM_F_NO_MULTI: a supplementary line carries 0x10000 (printed as 0x100) instead of 0x800 */
#define SSG_F_NOPAIRING 0x4    /* upstream MEM_F_NOPAIRING (-P): mate rescue, then every read on its own */
#define SSG_F_NO_RESCUE 0x20   /* upstream MEM_F_NO_RESCUE (-S) */
#define SSG_F_SOFTCLIP 0x200   /* upstream MEM_F_SOFTCLIP: supplementary lines keep soft clips and the whole SEQ / QUAL */

/* FM-index resident in HBM (upstream bwt_t + bntseq_t + pac).  The .bwt body is kept in its
 * on-disk interleaved form: one 64-byte block = 4 x u64 running counts + 8 x u32 (128 symbols),
 * i.e. one rank query = one 64-byte HBM line. */
typedef struct {
	const uint32_t *bwt;     /* device */
	const uint64_t *sa;      /* device; sa[0] = (uint64_t)-1 */
	const uint8_t  *pac;     /* device; 2-bit forward strand */
	const int64_t  *ctg_off; /* device; n_ctg contig offsets */
	const int32_t  *ctg_len; /* device */
	uint64_t primary, L2[5], seq_len;
	int64_t l_pac;
	int32_t n_ctg, sa_intv;
} ssg_index_view_t;

typedef struct { uint64_t x0, x1, x2, info; } ssg_intv_t;          /* upstream bwtintv_t */
typedef struct { int64_t rbeg; int32_t qbeg, len; int32_t score; int32_t next; } ssg_seed_t; /* mem_seed_t + chain link */

typedef struct {            /* upstream mem_chain_t (seeds as a linked list through ssg_seed_t.next) */
	int64_t pos;
	int32_t first_seed, last_seed, n, rid;
	int32_t w, kept, first, sec;   /* sec: tie rank reproducing the B-tree order among equal pos */
	int32_t left, right;           /* BST links while chaining */
	float frac_rep;
	int32_t _pad;
} ssg_chain_t;

typedef struct {            /* upstream mem_alnreg_t */
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp;
	float frac_rep;
	uint64_t hash;
} ssg_alnreg_t;

typedef struct { int32_t low, high, failed, _pad; double avg, std; } ssg_pestat_t; /* mem_pestat_t */

/* result of ksw_extend2 (a7) */
typedef struct { int32_t score, qle, tle, gtle, gscore, max_off; } ssg_ext_res_t;
/* one ksw_extend2 job for the stage-level entry point */
typedef struct { int32_t qoff, qlen, toff, tlen, w, end_bonus, zdrop, h0; } ssg_ext_job_t;

/* result of ksw_align2 (a10) */
typedef struct { int32_t score, te, qe, score2, te2, tb, qb; } ssg_kswr_t;
typedef struct { int32_t qoff, qlen, toff, tlen, xtra, _pad; } ssg_sw_job_t;

/* one ksw_global2 / bwa_gen_cigar2 job (a12) */
typedef struct { int32_t qoff, qlen, toff, tlen, w, _pad; } ssg_glb_job_t;

/* primary record of one read end as samblaster sees it: contig index (-1 unmapped), 1-based POS,
 * FLAG, leading / trailing clipped bases (S or H) and reference length of the CIGAR */
typedef struct { int32_t seq, pos, flag, lclip, rclip, ralen; } ssg_sbl_end_t;

/* one SAM line as samblaster reads it: contig index (-1 = '*'), 1-based POS, FLAG, MAPQ, and the CIGAR sums it derives --
 * leading / trailing clipped bases (S or H), query bases aligned (M I = X), reference bases covered (M D N = X) */
typedef struct { int32_t seq, pos, flag, mapq, lclip, rclip, qalen, ralen; } ssg_sbl_line_t;
/* samblaster's command-line switches (the reference passes --excludeDups --addMateTags --maxSplitCount --minNonOverlap,
 * bin/speedseq:439; maxUnmappedBases / minIndelSize keep upstream's defaults 50 / 50) */
typedef struct { int32_t exclude_dups, add_mate_tags, max_split_count, min_non_overlap, max_unmapped_bases, min_indel_size; } ssg_sbl_opt_t;

/* one SAM record to generate: a main record (primary / supplementary / unmapped) or an XA entry;
 * `owner` = region index (within the read) of the main record the entry belongs to */
typedef struct { int32_t read, reg, kind, owner, flag, mapq, _pad0, _pad1; } ssg_alnreq_t;
#define SSG_REQ_MAIN 0
#define SSG_REQ_XA   1

/* final alignment record (upstream mem_aln_t + the SAM fields mem_aln2sam derives) */
#define SSG_MAX_CIGAR 64
#define SSG_MAX_MD    320
typedef struct {
	int64_t pos;          /* 0-based on contig; -1 unmapped */
	int32_t rid, flag, mapq, NM, score, sub, n_cigar, is_rev, l_md, reg_idx, xa_cnt, _pad;
	uint32_t cigar[SSG_MAX_CIGAR];
	char md[SSG_MAX_MD];
} ssg_aln_t;


/* a pair of a batch whose lines can reach one of samblaster's side streams (a read with several main lines, or both ends mapped without the
 * proper-pair flag): where its records lie among the batch's BAM records (ssg_mem_process_fastq_bam) */
typedef struct { int64_t pair, first_rec, n_rec, byte_off, n_bytes; } ssg_bam_cand_t;

#endif
