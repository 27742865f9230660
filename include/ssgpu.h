/*
 * ssgpu.h -- C ABI of libssgpu.so, the MI355X-native `speedseq align` hot path
 * (BWA-MEM seed-and-extend + SAMBLASTER duplicate/discordant/splitter marking).
 *
 * This is the drop-in boundary below the reference's process-level plugin surface
 * (/root/reference/bin/speedseq.config:13-14 names the `bwa` and `samblaster` executables;
 * /root/reference/bin/speedseq:438-439 is the command line they must honour).  The reference's
 * src/bwa and src/samblaster are empty submodules, so each entry point cites the upstream
 * library function whose role it takes (upstream bwamem.h / ksw.h / bwt.h, SURVEY.md 8b last row)
 * and the SURVEY.md 8a row it implements.  Plain pointers and sizes only; the caller owns every
 * buffer it passes; all functions return 0 on success or a negative SSG_E* code; no globals
 * besides the HIP runtime's own state.  Host buffers are copied to HBM by the library; `*_dev`
 * variants take device pointers.
 */
#ifndef SSGPU_H
#define SSGPU_H
#include <stdint.h>
#include <stddef.h>
#include "../speedseq_amd/csrc/ssg_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SSG_OK         0
#define SSG_ENODEV   (-19)  /* no MI355X / HIP device: the library never falls back to the CPU */
#define SSG_ENOMEM   (-12)
#define SSG_EINVAL   (-22)
#define SSG_EIO      (-5)
#define SSG_EOVERFLOW (-75) /* an on-device capacity was exceeded; nothing was silently dropped */
#define SSG_EHIP     (-1000)

const char *ssg_version(void);
const char *ssg_backend(void);            /* "hip:gfx950" (or "emu" for the CPU test build) */
int ssg_device_count(void);
int ssg_set_device(int dev);
/* Several calls in flight on one device: a thread that calls ssg_set_lane(k), 0 < k < 8, runs everything it does afterwards on stream k of
 * its device with an arena of its own, so that one thread's uploads and downloads overlap another's kernels.  Lane 0 (the default)
 * is the default stream: one call at a time per device.  One thread per (device, lane); an index loaded on any lane serves all. */
int ssg_set_lane(int lane);
const char *ssg_last_error(void);

/* upstream mem_opt_init() (bwamem.c) -- SURVEY 8a defaults, Appendix B */
void ssg_mem_opt_init(ssg_mem_opt_t *opt);

/* ---- FM-index (upstream bwa_idx_load / bwa_idx_destroy, bwa.c; row a1) ---- */
typedef struct ssg_index ssg_index_t;
int ssg_index_load(const char *prefix, ssg_index_t **out);      /* reads prefix.{bwt,sa,pac,ann} into HBM */
/* The denser copy of the suffix array this library locates seeds through (every 4th row instead of the file's every 32nd) costs one
 * LF walk over the whole text when the index is loaded; it pays for itself after some tens of millions of reads.  A caller that does
 * not know how much input is coming loads with the file's density (defer_dense_sa != 0: same results, seed location walks further)
 * and calls ssg_index_densify() once the input has proved long -- between device calls, never while one runs on this index. */
int ssg_index_load2(const char *prefix, int defer_dense_sa, ssg_index_t **out);
int ssg_index_densify(ssg_index_t *idx);                         /* no-op when the index already is as dense as SSG_SA_INTV asks */
int ssg_index_densify_to(ssg_index_t *idx, int intv);             /* the same to every intv-th row (power of two below the current interval, else a no-op); the walk costs the same at any density */
/* (All index constructors fill the HBM copy of the suffix-array samples to every 4th row -- SSG_SA_INTV overrides -- with
 *  upstream's bwt_sa walk: 2 bytes of HBM per reference base, same locations, ~6x less work per located seed.) */
int ssg_index_from_arrays(const uint32_t *bwt, uint64_t bwt_words, uint64_t primary, const uint64_t L2[5],
                          const uint64_t *sa, uint64_t n_sa, int sa_intv,
                          const uint8_t *pac, int64_t l_pac,
                          int n_ctg, const int64_t *ctg_off, const int32_t *ctg_len, ssg_index_t **out);
/* upstream `bwa index` (bwtindex.c bwa_idx_build -> bntseq.c bns_fasta2bntseq + bwt_pac2bwt + bwt_bwtupdate_core + bwt_cal_sa;
 * the reference runs it at bin/speedseq:386-391 when any of the five index files is missing).  The suffix array
 * of fwd || revcomp(fwd) is built in HBM with 64-bit positions (no 2^31 / 2^32 limit); non-ACGT bases become lrand48() & 3
 * after srand48(11) and are recorded as .amb holes, as upstream does.  `fasta` may be gzip-compressed. */
int ssg_index_build_fasta(const char *fasta, ssg_index_t **out);
/* same from forward-strand nt4 codes (0..3, no holes) already resident in HBM; contigs are named "1".."n" until ssg_index_set_names */
int ssg_index_build_dev(const uint8_t *d_fwd, int64_t l_pac, int n_ctg, const int64_t *ctg_off, const int32_t *ctg_len, ssg_index_t **out);
/* upstream bns_dump + bwt_dump_bwt + bwt_dump_sa: prefix.{amb,ann,pac,bwt,sa} byte-identical to `bwa index` output (.sa interval 32) */
int ssg_index_save(const ssg_index_t *idx, const char *prefix);
void ssg_index_destroy(ssg_index_t *idx);
int64_t ssg_index_l_pac(const ssg_index_t *idx);
int ssg_index_n_ctg(const ssg_index_t *idx);

/* ---- stage-level batched entry points (host buffers) ---- */

/* upstream ksw_extend2() (ksw.c; row a7), one job per wavefront.  qbuf/tbuf hold nt4 codes. */
int ssg_extend_batch(const ssg_mem_opt_t *opt, int n_jobs, const ssg_ext_job_t *jobs,
                     const uint8_t *qbuf, size_t qbytes, const uint8_t *tbuf, size_t tbytes,
                     ssg_ext_res_t *res, uint64_t *cells);

/* upstream ksw_align2() (ksw.c; row a10), one job per wavefront. */
int ssg_align2_batch(const ssg_mem_opt_t *opt, int n_jobs, const ssg_sw_job_t *jobs,
                     const uint8_t *qbuf, size_t qbytes, const uint8_t *tbuf, size_t tbytes, ssg_kswr_t *res);

/* upstream ksw_global2() (ksw.c; row a12): score + CIGAR (ops packed len<<4|op, "MIDSH").
 * cigar: n_jobs x cap entries. */
int ssg_global_batch(const ssg_mem_opt_t *opt, int n_jobs, const ssg_glb_job_t *jobs,
                     const uint8_t *qbuf, size_t qbytes, const uint8_t *tbuf, size_t tbytes,
                     int32_t *score, int32_t *n_cigar, uint32_t *cigar, int cap);

/* upstream mem_collect_intv() (bwamem.c; rows a1-a2).  seq: concatenated nt4 codes, off[n_reads+1].
 * out_intv: n_reads x cap; out_n[r] = number of intervals (reads that overflow cap are re-run
 * internally with a larger capacity; SSG_EOVERFLOW if `cap` cannot hold a read's final list). */
int ssg_smem_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *seq, const int64_t *off,
                   int cap, ssg_intv_t *out_intv, int32_t *out_n);

/* Seeds of a batch of reads in upstream's visiting order (mem_chain(): for every interval of mem_collect_intv, every sampled occurrence:
 * bwt_sa() + bns_intv2rid(); rows a1-a3), before chaining.  seed_off[n_reads + 1]; *seeds and *rids (bns_intv2rid per seed, < 0 = a seed
 * upstream drops) are malloc'd by the library (ssg_free). */
int ssg_seeds_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *seq, const int64_t *off,
                    int64_t *seed_off, ssg_seed_t **seeds, int32_t **rids);

/* upstream ksw_extend2() through the LANE-per-extension kernel code the product's mem_chain2aln path runs (k_extlane.h; ssg_extend_batch
 * above goes through the wave-per-extension code): job i's target is jobs[i].tlen bases from doubled coordinate tpos[i] of the index's
 * 2-bit reference in direction dir (+1 / -1; jobs[i].toff is ignored), its query qbuf[qoff .. qoff + qlen).  qcap selects the kernel
 * instance (72 / 136 / 256 columns of LDS per lane; every qlen must fit).  Limits of the packed DP cells: h0 + qlen * a < 8191. */
int ssg_extend_lane_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_jobs, const ssg_ext_job_t *jobs, const int64_t *tpos, int dir,
                          const uint8_t *qbuf, size_t qbytes, int qcap, ssg_ext_res_t *res, uint64_t *cells);

/* upstream ksw_align2() as mate rescue runs it in the product path (row a10; k_mswlane.h): the forward pass of every job by the LANE kernel
 * (`lanes` = 1, 2 or 4 lanes per job, strips of 8 target rows in registers, strip boundaries in LDS), the reverse pass (KSW_XSTART) by the
 * wave code; a job the lane kernel does not take (KSW_XSTOP, a window above 8192 rows, scores beyond the packed cells: a > 15, b > 16,
 * qlen * a > 8190) goes through the wave code entirely.  Job i's target is jobs[i].tlen bases from doubled coordinate
 * tpos[i] of the index's 2-bit reference (jobs[i].toff is ignored), its query qbuf[qoff .. qoff + qlen).  from_lane[i] (may be NULL) = 1 when
 * the forward pass of job i came from the lane kernel. */
int ssg_align2_lane_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_jobs, const ssg_sw_job_t *jobs, const int64_t *tpos,
                          const uint8_t *qbuf, size_t qbytes, int lanes, ssg_kswr_t *res, int32_t *from_lane);

/* Test hook for the chaining kernels' weight sort (csrc/k_chainw.h wv_introsort_whi; upstream ks_introsort over chain weights, bwamem.c mem_chain_flt's
 * ks_introsort(mem_flt, ...) with its unstable tie order): sorts n <= 5120 words (w << 32 | id) by w, descending, once by the whole wave (out_wave) and
 * once by one lane running the textbook loops (out_lane); the two must be equal word for word. */
int ssg_dbg_chain_sort(const int64_t *keys, int n, int64_t *out_lane, int64_t *out_wave);

/* upstream mem_align1_core() (bwamem.c; rows a1-a8) for a batch of reads: SMEM -> SAL -> chain ->
 * filter -> extend -> sort/dedup/patch.  reg_off[n_reads+1] and regs (malloc'd by the library,
 * release with ssg_free) receive each read's mem_alnreg_t list in upstream order. */
int ssg_align1_batch(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *seq, const int64_t *off,
                     int64_t *reg_off, ssg_alnreg_t **regs, uint64_t stats[8]);

/* ---- the whole `bwa mem` paired-end hot path for a batch of pairs ----
 * upstream mem_process_seqs() (bwamem.c): worker1 = mem_align1_core per read, mem_pestat per
 * upstream batch, worker2 = mem_sam_pe per pair (rows a1-a12).
 *   seq/off     2*n_pairs reads (read1, read2 interleaved), nt4 codes, off[2*n_pairs+1]
 *   pair_batch  upstream batch index of every pair (insert-size statistics are per batch, row a9);
 *               batches are what `bwa mem -t N` would have formed (chunk_size*N bases, even count)
 *   id0         ordinal of the first pair in the whole input (upstream n_processed>>1; feeds the
 *               tie-breaking hash of mem_mark_primary_se / mem_pair)
 *   pes0        NULL to infer insert sizes, or 4 orientation entries as from `-I`
 * The result holds, per read, the SAM records to print (main records first, then XA entries). */
typedef struct ssg_pe_result ssg_pe_result_t;
int ssg_mem_process_pairs(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *seq, const int64_t *off,
                          const int32_t *pair_batch, int n_batches, int64_t id0, const ssg_pestat_t *pes0, ssg_pe_result_t **out);
/* Single-end reads (upstream mem_process_seqs without MEM_F_PE, bwamem.c: worker1 = mem_align1_core, worker2 = mem_mark_primary_se with
 * id = n_processed + i, then mem_reg2sam without a mate): n_reads reads, id0 = upstream's n_processed at the first of them.  The result
 * is read through the same accessors (req_off has n_reads + 1 entries, there is no insert-size model) and printed by ssg_sam_format_se. */
int ssg_mem_process_reads(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_reads, const uint8_t *seq, const int64_t *off, int64_t id0, ssg_pe_result_t **out);
void ssg_pe_result_free(ssg_pe_result_t *r);
/* optional warm-up: page-locks the host blocks the records of `n_calls` concurrent ssg_mem_process_pairs results of about `n_pairs`
 * pairs will land in (the first calls make them otherwise, ~0.2 s each); bin/bwa runs it next to the index load */
int ssg_pe_reserve(int n_pairs, int n_calls);
int64_t ssg_pe_n_req(const ssg_pe_result_t *r);
int ssg_pe_is_se(const ssg_pe_result_t *r);                         /* 1: made by ssg_mem_process_reads */
const int64_t *ssg_pe_req_off(const ssg_pe_result_t *r);          /* 2*n_pairs+1 offsets into req/alns */
const ssg_alnreq_t *ssg_pe_req(const ssg_pe_result_t *r);     /* kind 0 = SAM record, 1 = XA entry (owner = region) */
const ssg_aln_t *ssg_pe_alns(const ssg_pe_result_t *r);
const ssg_pestat_t *ssg_pe_pes(const ssg_pe_result_t *r);         /* n_batches x 4 (FF, FR, RF, RR) */
const uint64_t *ssg_pe_stats(const ssg_pe_result_t *r);           /* [0] seeds [1] extension cells [2] rescue cells [3] rescues [4] records */

/* upstream mem_aln2sam() (bwamem.c; row a13): SAM text for every read of the batch, in input order.
 * names/quals/comments: one C string per read (quals/comments entries may be NULL).  *sam is
 * malloc'd (release with ssg_free); sam_off[2*n_pairs+1] delimits each read's lines. */
int ssg_sam_format(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, int n_pairs,
                   const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals, const char *const *comments,
                   const char *rg_id, char **sam, int64_t *sam_off);
int ssg_sam_format_se(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, int n_reads,
                      const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals, const char *const *comments,
                      const char *rg_id, char **sam, int64_t *sam_off);   /* single-end results: sam_off[n_reads + 1] */
/* the same for a selection of the batch's pairs: sel[0..n_sel) are pair indices, sam_off[2*n_sel+1] */
int ssg_sam_format_sel(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, const int32_t *sel, int n_sel,
                       const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals, const char *const *comments,
                       const char *rg_id, char **sam, int64_t *sam_off);
/* the records of ssg_sam_format's lines as BAM (what `sambamba view -S -f bam`, the next stage of the reference's pipeline at
 * /root/reference/bin/speedseq:440, makes of that text: htslib-1.3.1 sam.c:835-1028 sam_parse1, sam.c:443-473 bam_write1): block_size-prefixed
 * records of every read in input order; bam_off[2*n_pairs+1] delimits each read's records.  Used by the fused plugin path,
 * where `bwa mem` hands binary records to samblaster / sambamba instead of SAM text.  *bam is malloc'd (ssg_free). */
int ssg_bam_format(const ssg_index_t *idx, const ssg_mem_opt_t *opt, const ssg_pe_result_t *res, int n_pairs,
                   const char *const *names, const uint8_t *seq, const int64_t *off, const char *const *quals,
                   const char *rg_id, uint8_t **bam, int64_t *bam_off);
/* ---- the same path with both text ends on the device (SURVEY 2.1 K1 + K11; rows a13, f2) ----
 * upstream bseq_read / kseq_read (htslib kseq.h:189-229) + mem_process_seqs + what `sambamba view -S -f bam` makes of mem_aln2sam's lines
 * (sam.c:835-1028 sam_parse1, sam.c:443-473 bam_write1) in ONE call: the FASTQ text of 2 * n_pairs plain four-line records goes to the device as it
 * is, in n_parts pieces (the two input files, say; page-locked pieces from ssg_host_alloc travel at bus speed) -- byte rec_off[r] of their
 * concatenation is the '@' of read r (read1, read2 interleaved; the caller's line scanner found them and checked the four-line form; the
 * device checks again; a record does not straddle two pieces) -- names, nt4 codes and qualities are taken from it there, and the block_size-prefixed BAM records of all reads come back
 * in input order, with the list of the pairs whose lines can reach one of samblaster's side streams under any options.  No comments (-C) on this
 * path.  pair_batch / n_batches / id0 / pes0 as for ssg_mem_process_pairs; rg_id: RG:Z value or NULL.  SSG_EINVAL with upstream's words when the
 * two reads of a pair carry different names. */
typedef struct ssg_pe_bam ssg_pe_bam_t;
int ssg_mem_process_fastq_bam(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *const *parts, const int64_t *part_bytes, int n_parts,
                              const int64_t *rec_off, const int32_t *pair_batch, int n_batches, int64_t id0, const ssg_pestat_t *pes0, const char *rg_id, ssg_pe_bam_t **out);
/* the same from parsed reads (codes, names, qualities as C strings; quals or quals[r] may be NULL): any FASTQ / FASTA the reader accepts */
int ssg_mem_process_pairs_bam(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *seq, const int64_t *off,
                              const char *const *names, const char *const *quals, const int32_t *pair_batch, int n_batches, int64_t id0,
                              const ssg_pestat_t *pes0, const char *rg_id, ssg_pe_bam_t **out);
void ssg_pe_bam_free(ssg_pe_bam_t *r);
const uint8_t *ssg_pe_bam_data(const ssg_pe_bam_t *r);            /* page-locked; ssg_pe_bam_bytes() bytes */
int64_t ssg_pe_bam_bytes(const ssg_pe_bam_t *r);
int64_t ssg_pe_bam_n_rec(const ssg_pe_bam_t *r);
int64_t ssg_pe_bam_n_cand(const ssg_pe_bam_t *r);
const ssg_bam_cand_t *ssg_pe_bam_cands(const ssg_pe_bam_t *r);    /* in pair order */
const ssg_pestat_t *ssg_pe_bam_pes(const ssg_pe_bam_t *r);        /* n_batches x 4 */
const uint64_t *ssg_pe_bam_stats(const ssg_pe_bam_t *r);          /* as ssg_pe_stats */
int ssg_index_set_names(ssg_index_t *idx, int n, const char *const *names);
const char *ssg_index_name(const ssg_index_t *idx, int i);
int32_t ssg_index_len(const ssg_index_t *idx, int i);

/* ---- SAMBLASTER duplicate marking (upstream samblaster.cpp markDups; row a14) ----
 * ends: 2*n_pairs primary records (read1, read2 per pair) in input order; dup[p] = 1 when an
 * earlier pair of the same call carries the same 5'-unclipped signature (first seen wins). */
int ssg_sbl_markdup(long n_pairs, const ssg_sbl_end_t *ends, uint8_t *dup);
/* streaming form: the signature table persists in HBM between calls (first-seen-wins over the whole
 * stream, as upstream's single pass) */
typedef struct ssg_sbl_state ssg_sbl_state_t;
ssg_sbl_state_t *ssg_sbl_state_new(void);
void ssg_sbl_state_free(ssg_sbl_state_t *st);
int ssg_sbl_markdup_stream(ssg_sbl_state_t *st, long n_pairs, const ssg_sbl_end_t *ends, uint8_t *dup);

/* upstream samblaster's per-block decisions on the device (samblaster.cpp block processing; rows a14-a17): `lines` are the SAM
 * lines of n_blocks name-grouped blocks (blk_off[n_blocks+1]) as numbers; on return line_bits[i] has SSG_SBL_DUP (OR 0x400 into
 * FLAG), SSG_SBL_DISC (copy to --discordantFile), SSG_SBL_SPLIT (copy to --splitterFile with _1/_2) and mate_line[i] is the
 * line whose CIGAR / MAPQ fill MC:Z / MQ:i (-1: none).  The duplicate set persists in `st` across calls (first seen wins
 * over the whole stream); st == NULL restricts it to this call. */
#define SSG_SBL_DUP   1
#define SSG_SBL_DISC  2
#define SSG_SBL_SPLIT 4
void ssg_sbl_opt_init(ssg_sbl_opt_t *o);
int ssg_sbl_process(ssg_sbl_state_t *st, const ssg_sbl_opt_t *o, long n_blocks, const int64_t *blk_off, const ssg_sbl_line_t *lines,
                    uint8_t *line_bits, int64_t *mate_line);
/* ssg_sbl_process in two halves, for several pipelines that share ONE duplicate set (rank mode: bin/speedseq-ranks).
 * samblaster keeps the first pair of every signature in input order (GregoryFaust/samblaster samblaster.cpp, the duplicate hash set; SURVEY 8a
 * a14): a pipeline extracts the two primary ends of every block (ends[2 * n_blocks]), the process that owns the set decides them with
 * ssg_sbl_markdup_stream in input order, and the lines are classified with those verdicts (dup[n_blocks]). */
int ssg_sbl_ends(long n_blocks, const int64_t *blk_off, const ssg_sbl_line_t *lines, ssg_sbl_end_t *ends);
int ssg_sbl_classify(const ssg_sbl_opt_t *o, long n_blocks, const int64_t *blk_off, const ssg_sbl_line_t *lines, const uint8_t *dup,
                     uint8_t *line_bits, int64_t *mate_line);

/* stable device radix sort of 64-bit keys, returned as a permutation: the coordinate sort of BAM records (samtools bam_sort.c:1607-1614
 * key tid<<32 | (pos+1)<<1 | reverse; ties keep input order) for the `sambamba sort` the reference runs at bin/speedseq:427 (row f1) */
int ssg_sort_u64_perm(const uint64_t *keys, int64_t n, uint32_t *perm);

/* ---- the measured hot path with device-resident inputs (bench.py) ----
 * d_seq / d_off / d_pair_batch are DEVICE pointers; aligned + duplicate-marked records stay in HBM.
 * summary: [0] records [1] duplicate pairs [2] seeds [3] extension cells [4] rescue cells [5] rescues */
int ssg_hotpath_dev(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, int max_len, const uint8_t *d_seq, const int64_t *d_off,
                    const int32_t *d_pair_batch, int n_batches, int64_t id0, uint64_t summary[8], uint8_t *dup_host);
/* Same, and the 5'-unclipped pair signatures (n_pairs x 3 uint64 in DEVICE memory; all ones = never a duplicate) that ranks
 * exchange for exact first-seen-wins duplicate marking over a sharded input (samblaster.cpp: the signature set is global). */
int ssg_hotpath_dev_sig(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, int max_len, const uint8_t *d_seq, const int64_t *d_off,
                        const int32_t *d_pair_batch, int n_batches, int64_t id0, uint64_t summary[8], uint8_t *dup_host, uint64_t *d_sig_out);

/* The whole path incl. samblaster's classification (rows a1-a12, a14-a17).  summary[0..7] as above, [8] discordant-stream lines,
 * [9] splitter-stream lines, [10] SAM lines.  sbl == NULL = the reference's switches (--excludeDups --addMateTags, bin/speedseq:439).
 * local_dedup = 0 + d_sig_out + keep: sharded input -- the caller exchanges the signatures between ranks, then classifies the
 * records still resident in HBM with the global verdicts (d_dup: DEVICE, one byte per pair; counts: dup pairs, discordant
 * lines, splitter lines, SAM lines). */
typedef struct ssg_dev_records ssg_dev_records_t;
int ssg_hotpath_dev_ex(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, int max_len, const uint8_t *d_seq, const int64_t *d_off,
                       const int32_t *d_pair_batch, int n_batches, int64_t id0, const ssg_sbl_opt_t *sbl, int local_dedup,
                       uint64_t summary[16], uint8_t *dup_host, uint64_t *d_sig_out, ssg_dev_records_t **keep);
int ssg_dev_records_classify(ssg_dev_records_t *r, const ssg_sbl_opt_t *sbl, const uint8_t *d_dup, uint64_t counts[4]);
/* the owner rank's side of that exchange (samblaster.cpp's set is global: the first pair of a signature in INPUT order survives): for n
 * signatures received from all ranks (DEVICE: d_sig n x 3 uint64, all ones = never a duplicate; d_ordinal = global input ordinal of the
 * pair, non-negative) d_dup[i] = 1 iff an element with the same signature and a smaller ordinal exists.  Same kernels as ssg_sbl_markdup. */
int ssg_markdup_sig_dev(long n, const uint64_t *d_sig, const int64_t *d_ordinal, uint8_t *d_dup);
void ssg_dev_records_free(ssg_dev_records_t *r);
/* for the coordinate-sorted merge across ranks (SURVEY 8e coupling 3): per SAM line of the kept records, in line order, the sort key
 * of samtools bam_sort.c:1607-1614 (tid<<32 | (pos+1)<<1 | reverse), the fixed-size device record and the side-stream bits,
 * written to DEVICE buffers of the caller (d_recs / d_bits may be NULL) */
int64_t ssg_dev_records_n_lines(const ssg_dev_records_t *r);
size_t ssg_dev_record_bytes(void);
int ssg_dev_records_export(const ssg_dev_records_t *r, uint64_t *d_keys, void *d_recs, uint8_t *d_bits);
/* the kept records as a host result (release with ssg_pe_result_free; prints through ssg_sam_format) and, per SAM line in line order
 * (ssg_dev_records_n_lines entries, HOST buffers, either may be NULL), samblaster's decisions: SSG_SBL_* bits and the mate line */
int ssg_dev_records_download(const ssg_dev_records_t *r, ssg_pe_result_t **res, uint8_t *line_bits, int64_t *mate_line);

/* FM-index from arrays already resident in HBM (not copied; the caller keeps them alive) */
int ssg_index_from_device(const uint32_t *d_bwt, uint64_t primary, const uint64_t L2[5], const uint64_t *d_sa, int sa_intv,
                          const uint8_t *d_pac, int64_t l_pac, int n_ctg, const int64_t *ctg_off, const int32_t *ctg_len, ssg_index_t **out);

/* per-kernel device time of the calls made since ssg_prof_reset() (HIP events on the launch stream).
 * ssg_prof_get: returns the number of kernels; name[i] / ms[i] / launches[i] for i < min(n, cap). */
void ssg_prof_enable(int on);
void ssg_prof_reset(void);
int ssg_prof_get(int cap, const char **name, double *ms, long *launches);

void ssg_free(void *p);

/* ---- the exchange step of the path over several GPUs (SURVEY 8e coupling 3: the coordinate-sorted merge; the reference ends in ONE OUT.bam,
 * /root/reference/bin/speedseq:441,491-495) ----
 * One process per GPU (bin/speedseq-ranks); every rank's `sambamba sort` owns a stretch of the genome and receives the records of that stretch from all ranks:
 * an all-to-all of variable-size blocks, here as grouped ncclSend / ncclRecv over RCCL (every xGMI link of a device at once; no ring).  ssg_coll_init: the
 * communicator over the calling process's device; rank 0 leaves the unique id in rendezvous_dir (a directory all ranks share).  Blocks are host buffers
 * (the records are made by host stages) staged through HBM.  SSG_ENODEV when there is no device or no RCCL: the caller's other transports take over. */
typedef struct ssg_coll ssg_coll_t;
int ssg_coll_available(void);   /* 1: a device is visible and RCCL loads (the ranks tell one another before any of them enters ssg_coll_init, which is collective) */
int ssg_coll_init(int rank, int world, const char *rendezvous_dir, ssg_coll_t **out);
int ssg_coll_alltoallv(ssg_coll_t *c, const void *const *send, const uint64_t *send_bytes, void *const *recv, const uint64_t *recv_bytes);   /* block q to / from rank q */
int ssg_coll_alltoall_u64(ssg_coll_t *c, const uint64_t *send, uint64_t *recv, int k);   /* k values to / from every rank */
void ssg_coll_destroy(ssg_coll_t *c);

/* ---- BGZF deflate on the device (row f1; htslib bgzf.c:298-342 is the format's writer in the reference) ----
 * Block b's payload is payload[cut[b] .. cut[b+1]) (<= 0xff00 bytes, as bgzf_write cuts them); its raw deflate stream (RFC 1951, one final
 * block; stored when it would not shrink) lands at out[out_off[b] .. out_off[b+1]).  The caller frames it: 18-byte BGZF header with the
 * block size, CRC-32 and ISIZE of the payload.  out_cap: bytes available at out (the sum of the payloads + 5 per block always suffices).
 * Host buffers; page-locked ones (ssg_host_alloc) travel at bus speed. */
int ssg_bgzf_deflate(const uint8_t *payload, const uint64_t *cut, long n_blocks, uint8_t *out, uint64_t out_cap, uint64_t *out_off);
void *ssg_host_alloc(size_t n);
void ssg_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
