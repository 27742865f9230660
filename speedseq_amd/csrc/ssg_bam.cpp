/*
 * ssg_bam.cpp -- the hot path with its two text ends on the device (k_bam.h; SURVEY.md 2.1 K1 + K11, row f2's fourth item): FASTQ text of plain
 * four-line records in, the BAM record bytes `sambamba view -S -f bam` would make of upstream's SAM lines out (htslib-1.3.1 sam.c:443-473,
 * 835-1028).  Between the two the records never leave HBM: text -> names / codes / qualities (ssg_k_fq_unpack, ssg_k_fq_codes) -> ssg_pe_core
 * (ssgpu_core.cpp: mem_process_seqs) -> sizes, offsets, bytes (ssg_k_bam_size, ssg_k_bam_write).  The host receives the bytes, their count, and
 * the list of pairs that can reach one of samblaster's side streams.  A translation unit of its own (seconds to compile).
 */
#include <algorithm>
#include <memory>
#include <string>
#include <vector>
#include "ssg_rt.h"
#include "k_bam.h"
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"
#include "ssg_pe_int.h"

SSG_ABI_FP_DEFINE(bam)
#define CHK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
#define CHKA(b) do { if (!(b).ok()) { ssg_err_msg = "device allocation failed: " #b; return SSG_ENOMEM; } } while (0)

struct ssg_pe_bam {
	int n_pairs = 0, n_batches = 0;
	hbuf<uint8_t> bam; int64_t bam_bytes = 0, n_rec = 0;
	hbuf<ssg_bam_cand_t> cands; int64_t n_cand = 0;
	std::vector<ssg_pestat_t> pes; uint64_t stats[8];
};

/* the contig names of the index in HBM (SA:Z / XA:Z name the contigs): made once per index, by whichever call comes first */
static int dev_names(const ssg_index *cix, const char **names, const int32_t **name_off)
{
	ssg_index *ix = const_cast<ssg_index*>(cix);
	std::lock_guard<std::mutex> l(ix->names_mu);
	if (!ix->d_names) {
		const int n = ix->v.n_ctg;
		std::vector<int32_t> off((size_t)n + 1, 0); std::string blob;
		for (int i = 0; i < n; ++i) { blob += i < (int)ix->names.size() ? ix->names[(size_t)i] : std::string("*"); off[(size_t)i + 1] = (int32_t)blob.size(); }
		char *d = (char*)rt_malloc(blob.size() + 1); int32_t *o = (int32_t*)rt_malloc(((size_t)n + 1) * 4);
		if (!d || !o) { rt_free(d); rt_free(o); ssg_err_msg = "device allocation failed: contig names"; return SSG_ENOMEM; }
		if (rt_h2d(d, blob.data(), blob.size()) || rt_h2d(o, off.data(), ((size_t)n + 1) * 4)) { rt_free(d); rt_free(o); return SSG_EHIP; }
		ix->d_names = d; ix->d_name_off = o;
	}
	*names = ix->d_names; *name_off = ix->d_name_off;
	return 0;
}

/* the records pe_core left in HBM -> BAM bytes on the host */
static int bam_stage(const ssg_index *idx, const ssg_mem_opt_t *opt, int n_pairs, const pe_dev_t &keep, const uint8_t *d_seq, const int64_t *d_off,
                     const uint8_t *d_text, const ssg_rdtext_t *d_rd, const char *rg_id, ssg_pe_bam *res)
{
	ssg_bam_ctx_t c;
	c.req = keep.req.p; c.alns = keep.alns.p; c.req_off = keep.req_off.p; c.seq = d_seq; c.off = d_off; c.text = d_text; c.rd = d_rd;
	CHK(dev_names(idx, &c.ctg_names, &c.ctg_name_off));
	const size_t l_rg = rg_id ? strlen(rg_id) : 0;
	dbuf<char> d_rg(l_rg + 1);
	CHKA(d_rg);
	if (l_rg) CHK(d_rg.up(rg_id, l_rg));
	c.rg_id = d_rg.p; c.l_rg = (int32_t)l_rg; c.softclip = (opt->flag & SSG_F_SOFTCLIP) ? 1 : 0;
	dbuf<int32_t> d_bytes(n_pairs), d_nrec(n_pairs), d_cand(n_pairs), d_err(1);
	dbuf<int64_t> d_boff((size_t)n_pairs + 1), d_roff((size_t)n_pairs + 1), d_coff((size_t)n_pairs + 1);
	CHKA(d_bytes); CHKA(d_nrec); CHKA(d_cand); CHKA(d_err); CHKA(d_boff); CHKA(d_roff); CHKA(d_coff);
	CHK(d_err.zero());
	SSG_LAUNCH(ssg_k_bam_size, (n_pairs + 63) / 64, 64, 0, c, (long)n_pairs, d_bytes.p, d_nrec.p, d_cand.p, d_err.p);
	CHK(ssg_dev_exclusive_scan(d_bytes.p, d_boff.p, n_pairs, &res->bam_bytes));
	CHK(ssg_dev_exclusive_scan(d_nrec.p, d_roff.p, n_pairs, &res->n_rec));
	CHK(ssg_dev_exclusive_scan(d_cand.p, d_coff.p, n_pairs, &res->n_cand));
	dbuf<uint8_t> d_bam((size_t)res->bam_bytes + 8); dbuf<ssg_bam_cand_t> d_cands((size_t)res->n_cand + 1);
	CHKA(d_bam); CHKA(d_cands);
	SSG_LAUNCH(ssg_k_bam_write, (n_pairs + 63) / 64, 64, 0, c, (long)n_pairs, (const int64_t*)d_boff.p, (const int64_t*)d_roff.p, (const int64_t*)d_coff.p, (const int32_t*)d_cand.p, d_bam.p, d_cands.p, d_err.p);
	int32_t err = 0;
	CHK(d_err.down(&err, 1));
	if (err) { ssg_err_msg = err == 1 ? "BAM records: a read without a record, or a pair of more than 2 GB" : "BAM records: the size pass and the write pass disagree"; return SSG_EOVERFLOW; }
	if (!res->bam.resize((size_t)res->bam_bytes + 8) || !res->cands.resize((size_t)res->n_cand + 1)) { ssg_err_msg = "host allocation failed: BAM records"; return SSG_ENOMEM; }
	CHK(d_bam.down(res->bam.data(), (size_t)res->bam_bytes)); CHK(d_cands.down(res->cands.data(), (size_t)res->n_cand));
	return 0;
}

static int check_args(int n_pairs, const int32_t *pair_batch, int n_batches, ssg_pe_bam_t **out)
{
	*out = 0;
	if (n_pairs <= 0 || n_batches <= 0) { ssg_err_msg = "ssg_mem_process_*_bam: empty input"; return SSG_EINVAL; }
	for (int p = 0; p < n_pairs; ++p) if (pair_batch[p] < 0 || pair_batch[p] >= n_batches) { ssg_err_msg = "pair_batch out of range"; return SSG_EINVAL; }
	return 0;
}

extern "C" {

int ssg_mem_process_fastq_bam(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *const *parts, const int64_t *part_bytes, int n_parts,
                              const int64_t *rec_off, const int32_t *pair_batch, int n_batches, int64_t id0, const ssg_pestat_t *pes0, const char *rg_id, ssg_pe_bam_t **out)
{
	CHK(ssg_need_device());
	CHK(check_args(n_pairs, pair_batch, n_batches, out));
	int64_t text_bytes = 0;
	for (int k = 0; k < n_parts; ++k) { if (part_bytes[k] < 0) { ssg_err_msg = "ssg_mem_process_fastq_bam: a part of negative length"; return SSG_EINVAL; } text_bytes += part_bytes[k]; }
	if (n_parts <= 0 || text_bytes <= 0) { ssg_err_msg = "ssg_mem_process_fastq_bam: no text"; return SSG_EINVAL; }
	const long n_reads = 2L * n_pairs;
	dbuf<uint8_t> d_text((size_t)text_bytes + 8); dbuf<int64_t> d_rec((size_t)n_reads), d_seq_at((size_t)n_reads), d_off((size_t)n_reads + 1);
	dbuf<ssg_rdtext_t> d_rd((size_t)n_reads); dbuf<int32_t> d_len((size_t)n_reads), d_err(4), d_pb(n_pairs);
	CHKA(d_text); CHKA(d_rec); CHKA(d_seq_at); CHKA(d_off); CHKA(d_rd); CHKA(d_len); CHKA(d_err); CHKA(d_pb);
	{ int64_t at = 0; for (int k = 0; k < n_parts; ++k) { CHK(rt_h2d(d_text.p + at, parts[k], (size_t)part_bytes[k])); at += part_bytes[k]; } }   /* the parts back to back: rec_off counts over their concatenation */
	CHK(d_rec.up(rec_off, (size_t)n_reads)); CHK(d_pb.up(pair_batch, n_pairs));
	{ const int32_t e0[4] = { 0, 0, 0x7fffffff, 0 }; CHK(d_err.up(e0, 4)); }
	SSG_LAUNCH(ssg_k_fq_unpack, (n_reads + 255) / 256, 256, 0, n_reads, (const uint8_t*)d_text.p, text_bytes, (const int64_t*)d_rec.p, d_rd.p, d_seq_at.p, d_len.p, d_err.p);
	SSG_LAUNCH(ssg_k_fq_pair_names, (n_pairs + 255) / 256, 256, 0, (long)n_pairs, (const uint8_t*)d_text.p, (const ssg_rdtext_t*)d_rd.p, d_err.p + 2);
	int64_t n_bases = 0;
	CHK(ssg_dev_exclusive_scan(d_len.p, d_off.p, n_reads, &n_bases));
	int32_t err[4];
	CHK(d_err.down(err, 4));
	if (err[0]) { ssg_err_msg = "ssg_mem_process_fastq_bam: the text is not one plain four-line record per offset (the caller's scanner and the device disagree)"; return SSG_EINVAL; }
	if (err[2] != 0x7fffffff) {   /* upstream mem_sam_pe's words, with the two names from the caller's text */
		const long p = err[2] - 1;
		auto name = [&](long r) {
			int64_t o = rec_off[r] + 1; int k = 0;
			while (k < n_parts && o >= part_bytes[k]) o -= part_bytes[k++];   /* a record lies within one part (the device found it plain) */
			if (k >= n_parts) return std::string("?");
			const uint8_t *s = parts[k] + o; size_t n = 0;
			while (o + (int64_t)n < part_bytes[k] && !(s[n] == ' ' || (s[n] >= 9 && s[n] <= 13))) ++n;
			if (n > 2 && s[n - 2] == '/' && s[n - 1] >= '0' && s[n - 1] <= '9') n -= 2;
			return std::string((const char*)s, n);
		};
		ssg_err_msg = "[mem_sam_pe] paired reads have different names: \"" + name(2 * p) + "\", \"" + name(2 * p + 1) + "\""; return SSG_EINVAL;
	}
	const int max_len = err[1];
	if (max_len > SSG_MAX_READ_LEN) { ssg_err_msg = "reads longer than 310 bases are outside this build's scope"; return SSG_EINVAL; }
	dbuf<uint8_t> d_seq((size_t)n_bases + 8);
	CHKA(d_seq);
	{ const long nw = std::min<long>(n_reads, 256L * 32); SSG_LAUNCH(ssg_k_fq_codes, (nw + 3) / 4, 256, 0, n_reads, (const uint8_t*)d_text.p, (const int64_t*)d_seq_at.p, (const int64_t*)d_off.p, d_seq.p); }
	std::unique_ptr<ssg_pe_bam> res(new ssg_pe_bam());
	res->n_pairs = n_pairs; res->n_batches = n_batches;
	ssg_pe_result pr; pe_dev_t keep;
	CHK(ssg_pe_core(idx, opt, n_pairs, d_seq.p, d_off.p, max_len, d_pb.p, n_batches, id0, pes0, &pr, &keep));
	res->pes = pr.pes; memcpy(res->stats, pr.stats, sizeof(res->stats));
	CHK(bam_stage(idx, opt, n_pairs, keep, d_seq.p, d_off.p, d_text.p, d_rd.p, rg_id, res.get()));
	ssg_prof_flush();
	*out = res.release();
	return 0;
}

int ssg_mem_process_pairs_bam(const ssg_index_t *idx, const ssg_mem_opt_t *opt, int n_pairs, const uint8_t *seq, const int64_t *off,
                              const char *const *names, const char *const *quals, const int32_t *pair_batch, int n_batches, int64_t id0,
                              const ssg_pestat_t *pes0, const char *rg_id, ssg_pe_bam_t **out)
{
	CHK(ssg_need_device());
	CHK(check_args(n_pairs, pair_batch, n_batches, out));
	const long n_reads = 2L * n_pairs;
	int max_len = 0; for (long r = 0; r < n_reads; ++r) max_len = std::max<int>(max_len, (int)(off[r + 1] - off[r]));
	if (max_len > SSG_MAX_READ_LEN) { ssg_err_msg = "reads longer than 310 bases are outside this build's scope"; return SSG_EINVAL; }
	/* names and qualities as one text: what ssg_k_fq_unpack would have found in the FASTQ */
	std::vector<ssg_rdtext_t> rd((size_t)n_reads); std::string text;
	text.reserve((size_t)off[n_reads] + (size_t)n_reads * 32);
	for (long r = 0; r < n_reads; ++r) {
		ssg_rdtext_t &t = rd[(size_t)r]; const size_t ln = strlen(names[r]);
		t.name_off = (int64_t)text.size(); t.l_name = (int32_t)ln; t._pad = 0; text.append(names[r], ln);
		if (quals && quals[r]) { t.qual_off = (int64_t)text.size(); text.append(quals[r], (size_t)(off[r + 1] - off[r])); } else t.qual_off = -1;
	}
	dbuf<uint8_t> d_text(text.size() + 8), d_seq((size_t)off[n_reads] + 8); dbuf<int64_t> d_off((size_t)n_reads + 1); dbuf<ssg_rdtext_t> d_rd((size_t)n_reads); dbuf<int32_t> d_pb(n_pairs);
	CHKA(d_text); CHKA(d_seq); CHKA(d_off); CHKA(d_rd); CHKA(d_pb);
	CHK(d_text.up((const uint8_t*)text.data(), text.size())); CHK(d_seq.up(seq, (size_t)off[n_reads])); CHK(d_off.up(off, (size_t)n_reads + 1)); CHK(d_rd.up(rd.data(), (size_t)n_reads)); CHK(d_pb.up(pair_batch, n_pairs));
	std::unique_ptr<ssg_pe_bam> res(new ssg_pe_bam());
	res->n_pairs = n_pairs; res->n_batches = n_batches;
	ssg_pe_result pr; pe_dev_t keep;
	CHK(ssg_pe_core(idx, opt, n_pairs, d_seq.p, d_off.p, max_len, d_pb.p, n_batches, id0, pes0, &pr, &keep));
	res->pes = pr.pes; memcpy(res->stats, pr.stats, sizeof(res->stats));
	CHK(bam_stage(idx, opt, n_pairs, keep, d_seq.p, d_off.p, d_text.p, d_rd.p, rg_id, res.get()));
	ssg_prof_flush();
	*out = res.release();
	return 0;
}

void ssg_pe_bam_free(ssg_pe_bam_t *r) { delete r; }
const uint8_t *ssg_pe_bam_data(const ssg_pe_bam_t *r) { return r->bam.data(); }
int64_t ssg_pe_bam_bytes(const ssg_pe_bam_t *r) { return r->bam_bytes; }
int64_t ssg_pe_bam_n_rec(const ssg_pe_bam_t *r) { return r->n_rec; }
int64_t ssg_pe_bam_n_cand(const ssg_pe_bam_t *r) { return r->n_cand; }
const ssg_bam_cand_t *ssg_pe_bam_cands(const ssg_pe_bam_t *r) { return r->cands.data(); }
const ssg_pestat_t *ssg_pe_bam_pes(const ssg_pe_bam_t *r) { return r->pes.data(); }
const uint64_t *ssg_pe_bam_stats(const ssg_pe_bam_t *r) { return r->stats; }

} /* extern "C" */
