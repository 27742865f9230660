/*
 * oracle/orc.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain scalar C restatement of the `speedseq align` hot path
 * (reference: /root/reference/bin/speedseq:438-439 invokes `$BWA mem | $SAMBLASTER`).
 *
 * PARITY UNPINNED for the alignment arithmetic: the reference tree's src/bwa and
 * src/samblaster are empty, un-vendored submodules (/root/reference/.gitmodules:4-6,16-18),
 * their pinned commits are unrecoverable and no golden SAM exists in the tree
 * (SURVEY.md section 8c).  The algorithms are therefore restated from the published
 * lh3/bwa (0.7.12-era) and GregoryFaust/samblaster behaviour as recalled in
 * SURVEY.md Appendix B/C; every function names the upstream function it restates.
 * PINNED: the FM-index on-disk format and contents (.pac/.bwt/.sa/.ann/.amb) are
 * checked byte-for-byte against /root/reference/example/data/ (tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- index (upstream bntseq.h / bwt.h) ---------- */
typedef struct { int64_t offset; int32_t len; int32_t n_ambs; uint32_t gi; int32_t is_alt; char *name, *anno; } orc_ann_t;
typedef struct { int64_t offset; int32_t len; char amb; } orc_amb_t;
typedef struct { int64_t l_pac; int32_t n_seqs; uint32_t seed; orc_ann_t *anns; int32_t n_holes; orc_amb_t *ambs; } orc_bns_t;
typedef struct {
	uint64_t primary, L2[5], seq_len, bwt_size; /* bwt_size in u32 words incl. occ checkpoints */
	uint32_t *bwt;
	int sa_intv; uint64_t n_sa; uint64_t *sa;
} orc_bwt_t;
typedef struct { orc_bwt_t *bwt; orc_bns_t *bns; uint8_t *pac; } orc_idx_t;

typedef struct { uint64_t x[3], info; } orc_intv_t; /* upstream bwtintv_t */

orc_idx_t *orc_idx_load(const char *prefix);
orc_idx_t *orc_idx_build_fasta(const char *fasta);           /* `bwa index` restatement (small refs) */
orc_idx_t *orc_idx_build_mem(int n_seqs, const char **names, const char **seqs); /* ASCII seqs */
int  orc_idx_save(const orc_idx_t *idx, const char *prefix);
void orc_idx_destroy(orc_idx_t *idx);

uint64_t orc_bwt_occ(const orc_bwt_t *bwt, uint64_t k, int c);
void orc_bwt_occ4(const orc_bwt_t *bwt, uint64_t k, uint64_t cnt[4]);
void orc_bwt_extend(const orc_bwt_t *bwt, const orc_intv_t *ik, orc_intv_t ok[4], int is_back);
uint64_t orc_bwt_sa(const orc_bwt_t *bwt, uint64_t k);
int  orc_bns_pos2rid(const orc_bns_t *bns, int64_t pos_f);
int  orc_bns_intv2rid(const orc_bns_t *bns, int64_t rb, int64_t re);
int64_t orc_bns_depos(const orc_bns_t *bns, int64_t pos, int *is_rev);
/* base at doubled-coordinate p in [0,2*l_pac) */
static inline int orc_pac_get(const uint8_t *pac, int64_t l) { return pac[l>>2] >> ((~l&3)<<1) & 3; }
static inline int orc_ref_base(const uint8_t *pac, int64_t l_pac, int64_t p) { return p < l_pac ? orc_pac_get(pac, p) : 3 - orc_pac_get(pac, (l_pac<<1) - 1 - p); }

/* ---------- options (upstream mem_opt_t, bwamem.h) ---------- */
#define ORC_F_NO_MULTI 0x10    /* upstream MEM_F_NO_MULTI (bwa mem -M) */
#define ORC_F_SOFTCLIP 0x200   /* upstream MEM_F_SOFTCLIP (bwa mem -Y) */
#define ORC_F_NOPAIRING 0x4    /* upstream MEM_F_NOPAIRING (bwa mem -P): mate rescue only, no pairing of the hits */
#define ORC_F_NO_RESCUE 0x20   /* upstream MEM_F_NO_RESCUE (bwa mem -S) */
typedef struct {
	int a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, pen_clip5, pen_clip3, w, zdrop;
	uint64_t max_mem_intv;
	int T, flag, min_seed_len, min_chain_weight, max_chain_extend;
	float split_factor; int split_width, max_occ, max_chain_gap;
	int n_threads, chunk_size;
	float mask_level, drop_ratio, XA_drop_ratio, mask_level_redun, mapQ_coef_len;
	int mapQ_coef_fac, max_ins, max_matesw, max_XA_hits, max_XA_hits_alt;
	int8_t mat[25];
} orc_opt_t;
void orc_opt_init(orc_opt_t *o);
void orc_fill_scmat(orc_opt_t *o);   /* after a change of a / b */

/* ---------- Smith-Waterman (upstream ksw.c) ---------- */
int orc_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                    int *qle, int *tle, int *gtle, int *gscore, int *max_off);
int orc_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t **cigar);
typedef struct { int score, te, qe, score2, te2, tb, qb; } orc_kswr_t;
#define ORC_KSW_XBYTE  0x10000
#define ORC_KSW_XSTOP  0x20000
#define ORC_KSW_XSUBO  0x40000
#define ORC_KSW_XSTART 0x80000
orc_kswr_t orc_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                          int o_del, int e_del, int o_ins, int e_ins, int xtra);
/* work counters: per thread -- a shared counter bumped in the inner loops makes the worker threads fight over one cache
 * line -- and folded into the totals by orc_cnt_flush() when a worker is done */
extern __thread uint64_t orc_cnt_cells __attribute__((tls_model("initial-exec")));                              /* SW cells */
extern __thread uint64_t orc_cnt_extend __attribute__((tls_model("initial-exec"))), orc_cnt_lf __attribute__((tls_model("initial-exec"))), orc_cnt_sa __attribute__((tls_model("initial-exec")));     /* FM-index work */
extern uint64_t orc_tot_cells, orc_tot_extend, orc_tot_lf, orc_tot_sa;
void orc_cnt_flush(void);

/* ---------- seeding / chaining / extension (upstream bwamem.c) ---------- */
typedef struct { int64_t rbeg; int32_t qbeg, len, score; } orc_seed_t;
typedef struct { int n, m, first, rid; uint32_t w:29, kept:2, is_alt:1; float frac_rep; int64_t pos; orc_seed_t *seeds; } orc_chain_t;
typedef struct { size_t n, m; orc_chain_t *a; } orc_chain_v;
typedef struct { size_t n, m; orc_intv_t *a; } orc_intv_v;

typedef struct {
	int64_t rb, re; int qb, qe; int rid; int score; int truesc; int sub; int alt_sc; int csub; int sub_n;
	int w; int seedcov; int secondary; int secondary_all; int seedlen0;
	int n_comp:30, is_alt:2; float frac_rep; uint64_t hash;
} orc_alnreg_t;
typedef struct { size_t n, m; orc_alnreg_t *a; } orc_alnreg_v;

int  orc_smem1(const orc_bwt_t *bwt, int len, const uint8_t *q, int x, int min_intv, uint64_t max_intv, orc_intv_v *mem, orc_intv_v tmp[2]);
int  orc_seed_strategy1(const orc_bwt_t *bwt, int len, const uint8_t *q, int x, int min_len, int max_intv, orc_intv_t *mem);
void orc_collect_intv(const orc_opt_t *opt, const orc_bwt_t *bwt, int len, const uint8_t *seq, orc_intv_v *mem);
orc_chain_v orc_mem_chain(const orc_opt_t *opt, const orc_idx_t *idx, int len, const uint8_t *seq);
/* the container of mem_chain: 0 = the position-sorted array with the single-leaf semantics of klib's B-tree (what the HIP kernels restate), 1 = the B-tree itself
 * (orc_mem.c; per thread).  orc_chain_exposure: one read through both; 1 when the chain lists differ */
void orc_set_chain_container(int kbtree);
int  orc_chain_exposure(const orc_opt_t *opt, const orc_idx_t *idx, int len, const uint8_t *seq, int *n_chains, int *dup);
int  orc_mem_chain_flt(const orc_opt_t *opt, int n_chn, orc_chain_t *a);
void orc_mem_chain2aln(const orc_opt_t *opt, const orc_idx_t *idx, int l_query, const uint8_t *query, const orc_chain_t *c, orc_alnreg_v *av);
int  orc_mem_sort_dedup_patch(const orc_opt_t *opt, const orc_idx_t *idx, uint8_t *query, int n, orc_alnreg_t *a);
orc_alnreg_v orc_mem_align1_core(const orc_opt_t *opt, const orc_idx_t *idx, int l_seq, uint8_t *seq /* nt4 codes */);

/* ---------- pairing / SAM (upstream bwamem_pair.c, bwamem.c) ---------- */
typedef struct { int low, high, failed; double avg, std; } orc_pestat_t;
typedef struct {
	int64_t pos; int rid; int flag; uint32_t is_rev:1, is_alt:1, mapq:8, NM:22;
	int n_cigar; uint32_t *cigar; /* MD string follows cigar ops */
	char *XA; int score, sub, alt_sc;
} orc_aln_t;
typedef struct { int l_seq; char *name, *comment, *qual; uint8_t *seq /* nt4 */; char *sam; } orc_read_t;

void orc_mem_pestat(const orc_opt_t *opt, int64_t l_pac, int n, const orc_alnreg_v *regs, orc_pestat_t pes[4]);
int  orc_mem_matesw(const orc_opt_t *opt, const orc_idx_t *idx, const orc_pestat_t pes[4], const orc_alnreg_t *a, int l_ms, const uint8_t *ms, orc_alnreg_v *ma);
int  orc_mem_mark_primary_se(const orc_opt_t *opt, int n, orc_alnreg_t *a, int64_t id);
int  orc_mem_approx_mapq_se(const orc_opt_t *opt, const orc_alnreg_t *a);
int  orc_mem_pair(const orc_opt_t *opt, const orc_idx_t *idx, const orc_pestat_t pes[4], const orc_alnreg_v a[2], int id, int *sub, int *n_sub, int z[2], int n_pri[2]);
orc_aln_t orc_mem_reg2aln(const orc_opt_t *opt, const orc_idx_t *idx, int l_query, const uint8_t *query, const orc_alnreg_t *ar);
int  orc_mem_sam_pe(const orc_opt_t *opt, const orc_idx_t *idx, const orc_pestat_t pes[4], uint64_t id, orc_read_t s[2], orc_alnreg_v a[2], const char *rg_id);
uint32_t *orc_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac,
                         int l_query, uint8_t *query, int64_t rb, int64_t re, int *score, int *n_cigar, int *NM);

/* process one upstream "batch" (mem_process_seqs): n reads (n even, interleaved pairs); fills s[i].sam */
void orc_mem_process_pairs(const orc_opt_t *opt, const orc_idx_t *idx, int64_t n_processed, int n, orc_read_t *s,
                           const orc_pestat_t *pes0, const char *rg_id, orc_pestat_t pes_out[4], int n_threads);
void orc_mem_process_reads(const orc_opt_t *opt, const orc_idx_t *idx, int64_t n_processed, int n, orc_read_t *s, const char *rg_id, int n_threads);
/* upstream bwa_print_sam_hdr: @SQ lines + optional @RG + @PG */
char *orc_sam_header(const orc_idx_t *idx, const char *rg_line, const char *pg_cl);

/* ---------- generic introsort with klib tie behaviour (htslib/ksort.h:178-229) ---------- */
typedef int (*orc_lt_f)(const void *a, const void *b);
void orc_introsort(void *base, size_t n, size_t sz, orc_lt_f lt);
void orc_introsort_u64(size_t n, uint64_t *a);

/* ---------- samblaster restatement (upstream samblaster.cpp) ---------- */
typedef struct {
	int exclude_dups, add_mate_tags, max_split_count, min_non_overlap, max_unmapped_bases, min_indel_size;
} orc_sbl_opt_t;
void orc_sbl_opt_init(orc_sbl_opt_t *o);
/* stream SAM from `in` to `out`; splitter/discordant may be NULL. returns 0 on success. stats[0]=pairs,[1]=dups */
int orc_samblaster(const orc_sbl_opt_t *o, FILE *in, FILE *out, FILE *splitter, FILE *discordant, uint64_t stats[4]);

#ifdef __cplusplus
}
#endif
#endif
