/*
 * k_sbl.h -- SAMBLASTER on the device (upstream GregoryFaust/samblaster samblaster.cpp as the reference
 * invokes it at /root/reference/bin/speedseq:439).
 *
 * Input is the name-grouped record stream as numbers: one ssg_sbl_line_t per SAM line (contig, POS, FLAG,
 * MAPQ and the CIGAR sums samblaster derives), blocks = lines of one QNAME.  One lane per block:
 *   ssg_k_sbl_ends      primaries of the block -> the two 5' ends the duplicate signature is built from (a14)
 *   ssg_k_sbl_classify  per line: duplicate bit (a14), mate line for MC/MQ (a15), discordant (a16) and
 *                       splitter (a17) side-stream bits -- the rules of oracle/orc_samblaster.c, lane for line.
 * The duplicate set of everything seen so far is an open-addressing hash table in HBM keyed by the 64-bit
 * signature hash (the "radix bucket" of the north star): a chunk is sorted by hash to settle first-seen-wins
 * among its own pairs, probed against the table, and its new signatures are inserted -- O(1) HBM lines per
 * pair however long the stream is.
 */
#ifndef SSG_K_SBL_H
#define SSG_K_SBL_H
#include "k_misc.h"

#ifndef SSG_SBL_DUP
#define SSG_SBL_DUP   1   /* OR 0x400 into the line's FLAG */
#define SSG_SBL_DISC  2   /* line goes to --discordantFile */
#define SSG_SBL_SPLIT 4   /* line goes to --splitterFile (QNAME + _1 / _2) */
#endif
#define SSG_SBL_MAX_SPLIT 16

/* primaries of block b (first 0x40 / first 0x80 line without 0x100 | 0x800) -> ends[2b], ends[2b+1]; prim[2b], prim[2b+1] = line index or -1 */
__global__ void ssg_k_sbl_ends(long n_blocks, const int64_t *blk_off, const ssg_sbl_line_t *lines, ssg_sbl_end_t *ends, int64_t *prim)
{
	const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	int64_t r1 = -1, r2 = -1;
	for (int64_t i = blk_off[b]; i < blk_off[b + 1]; ++i) {
		const int f = lines[i].flag;
		if (f & (0x100 | 0x800)) continue;
		if ((f & 0x40) && r1 < 0) r1 = i;
		else if ((f & 0x80) && r2 < 0) r2 = i;
	}
	prim[2 * b] = r1; prim[2 * b + 1] = r2;
	for (int e = 0; e < 2; ++e) {
		ssg_sbl_end_t x;
		if (r1 >= 0 && r2 >= 0) {
			const ssg_sbl_line_t l = lines[e ? r2 : r1];
			x.seq = ((l.flag & 0x4) || l.seq < 0) ? -1 : l.seq; x.pos = l.pos; x.flag = l.flag | (x.seq < 0 ? 0x4 : 0);
			x.lclip = l.lclip; x.rclip = l.rclip; x.ralen = l.ralen;
		} else { x.seq = -1; x.pos = 0; x.flag = 0x4; x.lclip = x.rclip = x.ralen = 0; }   /* unpaired block: never a duplicate */
		ends[2 * b + e] = x;
	}
}

/* ---- the persistent duplicate set: open addressing, linear probing; slot hash 0 = empty ---- */
SSG_DEVFN uint64_t ssg_sbl_slot_hash(uint64_t h) { return h ? h : 1; }
SSG_DEVFN bool ssg_sig_eq(const ssg_sig_t &a, const ssg_sig_t &b) { return (a.k0 == b.k0) & (a.k1 == b.k1) & (a.k2 == b.k2); }
SSG_DEVFN bool ssg_sig_never(const ssg_sig_t &s) { return s.k0 == ~0ull && s.k1 == ~0ull && s.k2 == ~0ull; }

SSG_DEVFN bool ssg_sbl_table_has(const uint64_t *th, const ssg_sig_t *ts, uint64_t mask, uint64_t h, const ssg_sig_t &s)
{
	h = ssg_sbl_slot_hash(h);
	for (uint64_t q = h & mask; ; q = (q + 1) & mask) {
		const uint64_t v = th[q];
		if (v == 0) return false;
		if (v == h && ssg_sig_eq(ts[q], s)) return true;
	}
}
SSG_DEVFN void ssg_sbl_table_put(uint64_t *th, ssg_sig_t *ts, uint64_t mask, uint64_t h, const ssg_sig_t &s)
{	/* the caller guarantees the signature is not in the table and that no other lane inserts the same signature */
	h = ssg_sbl_slot_hash(h);
	for (uint64_t q = h & mask; ; q = (q + 1) & mask) {
		if (th[q] == 0 && atomicCAS((unsigned long long*)&th[q], 0ull, (unsigned long long)h) == 0ull) { ts[q] = s; return; }
	}
}

/* After a STABLE sort of the chunk's (hash, ordinal) by hash: element i is a duplicate iff an earlier element of its
 * equal-hash run carries the identical signature (smaller ordinal = seen first) or the signature is in the table.
 * fresh[o] = 1 for the first occurrence of a signature that the table does not hold yet (to be inserted). */
__global__ void ssg_k_sbl_markdup(long n, const uint64_t *hash_sorted, const uint32_t *ord_sorted, const ssg_sig_t *sig,
                                  const uint64_t *th, const ssg_sig_t *ts, uint64_t mask, uint8_t *dup, uint8_t *fresh)
{
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t h = hash_sorted[i]; const uint32_t o = ord_sorted[i];
	const ssg_sig_t s = sig[o];
	int d = 0, f = 0;
	if (!ssg_sig_never(s)) {
		for (long j = i - 1; j >= 0 && hash_sorted[j] == h && !d; --j) d = ssg_sig_eq(sig[ord_sorted[j]], s);
		if (!d) { if (th && ssg_sbl_table_has(th, ts, mask, h, s)) d = 1; else f = 1; }
	}
	dup[o] = (uint8_t)d;
	if (fresh) fresh[o] = (uint8_t)f;
}
__global__ void ssg_k_sbl_insert(long n, const uint64_t *hash, const ssg_sig_t *sig, const uint8_t *fresh, uint64_t *th, ssg_sig_t *ts, uint64_t mask)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p < n && fresh[p]) ssg_sbl_table_put(th, ts, mask, hash[p], sig[p]);
}
/* growth: every occupied slot of the old table into the new one */
__global__ void ssg_k_sbl_rehash(uint64_t old_slots, const uint64_t *oh, const ssg_sig_t *os, uint64_t *th, ssg_sig_t *ts, uint64_t mask)
{
	const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q < old_slots && oh[q]) ssg_sbl_table_put(th, ts, mask, oh[q], os[q]);
}

/* ---- a15-a17 ---- */
SSG_DEVFN void ssg_sbl_mark_splitters(const ssg_sbl_opt_t &o, const ssg_sbl_line_t *lines, int64_t b0, int64_t b1, int mask, uint8_t *out)
{	/* upstream's split-read test for one read of the block (oracle/orc_samblaster.c mark_splitters) */
	int64_t arr[SSG_SBL_MAX_SPLIT]; int sq[SSG_SBL_MAX_SPLIT], eq[SSG_SBL_MAX_SPLIT]; int cnt = 0;
	for (int64_t i = b0; i < b1; ++i) if (lines[i].flag & mask) { if (cnt == SSG_SBL_MAX_SPLIT) return; arr[cnt++] = i; }
	if (cnt < 2 || cnt > o.max_split_count) return;
	for (int k = 0; k < cnt; ++k) {
		const ssg_sbl_line_t l = lines[arr[k]];
		if ((l.flag & 0x4) || l.seq < 0) return;
		sq[k] = (l.flag & 0x10) ? l.rclip : l.lclip; eq[k] = sq[k] + l.qalen - 1;
	}
	for (int k = 1; k < cnt; ++k) {   /* stable insertion sort by query start */
		const int64_t t = arr[k]; const int ts = sq[k], te = eq[k]; int j = k;
		while (j > 0 && sq[j - 1] > ts) { arr[j] = arr[j - 1]; sq[j] = sq[j - 1]; eq[j] = eq[j - 1]; --j; }
		arr[j] = t; sq[j] = ts; eq[j] = te;
	}
	for (int k = 1; k < cnt; ++k) {
		const ssg_sbl_line_t L = lines[arr[k - 1]], R = lines[arr[k]];
		const int lo = sq[k - 1] > sq[k] ? sq[k - 1] : sq[k], hi = eq[k - 1] < eq[k] ? eq[k - 1] : eq[k];
		int overlap = 1 + hi - lo; if (overlap < 0) overlap = 0;
		const int alen1 = 1 + eq[k - 1] - sq[k - 1], alen2 = 1 + eq[k] - sq[k];
		const int mno = (alen1 < alen2 ? alen1 : alen2) - overlap;
		const int desert = sq[k] - eq[k - 1] - 1; int ok = 1;
		if (mno < o.min_non_overlap) ok = 0;
		else if (L.seq == R.seq && (L.flag & 0x10) == (R.flag & 0x10)) {
			int64_t ld, rd, ins;
			if (!(L.flag & 0x10)) { ld = (int64_t)L.pos - sq[k - 1]; rd = (int64_t)R.pos - sq[k]; ins = rd - ld; }
			else { ld = (int64_t)L.pos + L.ralen - 1 + sq[k - 1]; rd = (int64_t)R.pos + R.ralen - 1 + sq[k]; ins = ld - rd; }
			if (desert > 0 && desert - (ins > 0 ? ins : 0) > o.max_unmapped_bases) ok = 0;
			if ((ins < 0 ? -ins : ins) < o.min_indel_size) ok = 0;
		} else if (desert > o.max_unmapped_bases) ok = 0;
		if (ok) { out[arr[k - 1]] |= SSG_SBL_SPLIT; out[arr[k]] |= SSG_SBL_SPLIT; }
	}
}

__global__ void ssg_k_sbl_classify(ssg_sbl_opt_t o, long n_blocks, const int64_t *blk_off, const ssg_sbl_line_t *lines, const int64_t *prim,
                                   const uint8_t *dup, uint8_t *out, int64_t *mate_line)
{
	const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	const int64_t b0 = blk_off[b], b1 = blk_off[b + 1], r1 = prim[2 * b], r2 = prim[2 * b + 1];
	const int d = dup[b] && r1 >= 0 && r2 >= 0;
	for (int64_t i = b0; i < b1; ++i) {
		out[i] = d ? SSG_SBL_DUP : 0;
		const int f = lines[i].flag;
		mate_line[i] = (r1 >= 0 && r2 >= 0) ? ((f & 0x40) ? r2 : (f & 0x80) ? r1 : -1) : -1;   /* MC/MQ come from the mate's primary line */
	}
	if ((d && o.exclude_dups) || r1 < 0 || r2 < 0) return;
	const ssg_sbl_line_t p1 = lines[r1], p2 = lines[r2];
	if (!(p1.flag & 0x4) && !(p2.flag & 0x4) && p1.seq >= 0 && p2.seq >= 0 && !(p1.flag & 0x2)) { out[r1] |= SSG_SBL_DISC; out[r2] |= SSG_SBL_DISC; }
	ssg_sbl_mark_splitters(o, lines, b0, b1, 0x40, out);
	ssg_sbl_mark_splitters(o, lines, b0, b1, 0x80, out);
}

/* coordinate-sort key of every SAM line (samtools bam_sort.c:1607-1614: tid<<32 | (pos+1)<<1 | reverse, unplaced lines last) and its
 * fixed-size device record, gathered in line order: what a rank hands to the range exchange of the sorted merge (SURVEY.md 8e coupling 3) */
__global__ void ssg_k_sbl_export(int64_t n_lines, const ssg_sbl_line_t *lines, const int64_t *line_req, const ssg_aln_t *alns, const uint8_t *bits,
                                 uint64_t *key, ssg_aln_t *recs, uint8_t *bits_out)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_lines) return;
	const ssg_sbl_line_t l = lines[i];
	const uint64_t tid = l.seq < 0 ? 0xffffffffull : (uint64_t)(uint32_t)l.seq;
	const uint32_t pos1 = l.seq < 0 ? 0u : (uint32_t)l.pos;          /* BAM pos + 1 = SAM POS */
	key[i] = tid << 32 | (uint64_t)(pos1 << 1) | ((l.flag & 0x10) ? 1u : 0u);
	if (recs) recs[i] = alns[line_req[i]];
	if (bits_out) bits_out[i] = bits[i];
}

/* c[0] += duplicate pairs, c[1] += discordant-stream lines, c[2] += splitter-stream lines; one atomic per wave and counter */
__global__ void ssg_k_sbl_count_bits(int64_t n_lines, const uint8_t *bits, long n_pairs, const uint8_t *dup, unsigned int *c)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int b = i < n_lines ? bits[i] : 0, d = i < n_pairs ? dup[i] : 0;
	const unsigned long long b0 = wv_ballot(d != 0), b1 = wv_ballot(b & SSG_SBL_DISC), b2 = wv_ballot(b & SSG_SBL_SPLIT);
	if (wv_lane() == 0) {
		if (b0) atomicAdd(&c[0], (unsigned)__popcll(b0));
		if (b1) atomicAdd(&c[1], (unsigned)__popcll(b1));
		if (b2) atomicAdd(&c[2], (unsigned)__popcll(b2));
	}
}

/* fused path: the SAM lines a pair's device records will print as (main requests only; XA entries are tags, not lines).
 * n_line[p] is counted first (ssg_k_sbl_count_lines), offsets by prefix sum, then the lines are filled. */
__global__ void ssg_k_sbl_count_lines(long n_pairs, const int64_t *req_off, const ssg_alnreq_t *req, int32_t *n_line)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	int c = 0;
	for (int64_t g = req_off[2 * p]; g < req_off[2 * p + 2]; ++g) c += req[g].kind == SSG_REQ_MAIN;
	n_line[p] = c;
}
/* FLAG exactly as mem_aln2sam prints it (sam_format.cpp aln2sam): paired, own / mate unmapped and strand bits added */
__global__ void ssg_k_sbl_lines_from_alns(long n_pairs, const int64_t *req_off, const ssg_alnreq_t *req, const ssg_aln_t *alns, const int64_t *line_off,
                                          ssg_sbl_line_t *lines, int64_t *line_req)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	int64_t w = line_off[p];
	for (int e = 0; e < 2; ++e) {
		const int64_t g0 = req_off[2 * p + e], g1 = req_off[2 * p + e + 1], m0 = req_off[2 * p + (e ^ 1)];
		const ssg_aln_t &mh = alns[m0];     /* the mate's first record: what this read's lines print against */
		int which = 0;
		for (int64_t g = g0; g < g1; ++g) {
			if (req[g].kind != SSG_REQ_MAIN) continue;
			const ssg_aln_t &a = alns[g];
			int flag = a.flag | 0x1, rid = a.rid, is_rev = a.is_rev, n_cigar = a.n_cigar; int64_t pos = a.pos;
			int m_rid = mh.rid, m_rev = mh.is_rev;
			flag |= rid < 0 ? 0x4 : 0;
			flag |= m_rid < 0 ? 0x8 : 0;
			if (rid < 0 && m_rid >= 0) { rid = m_rid; pos = mh.pos; is_rev = m_rev; n_cigar = 0; }
			if (m_rid < 0 && a.rid >= 0) m_rev = a.is_rev;
			flag |= is_rev ? 0x10 : 0;
			flag |= m_rev ? 0x20 : 0;
			flag = (flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0);
			ssg_sbl_line_t l; l.seq = rid; l.pos = (int32_t)(pos + 1); l.flag = flag; l.mapq = rid >= 0 ? a.mapq : 0;
			l.lclip = l.rclip = l.qalen = l.ralen = 0;
			int first = 1, rc = 0;
			for (int k = 0; k < n_cigar; ++k) {
				const int op = a.cigar[k] & 0xf, len = (int)(a.cigar[k] >> 4);
				if (op == 3 || op == 4) { if (first) l.lclip += len; rc += len; }
				else { first = 0; rc = 0; if (op == 0) { l.qalen += len; l.ralen += len; } else if (op == 1) l.qalen += len; else if (op == 2) l.ralen += len; }
			}
			l.rclip = (l.qalen + l.ralen) ? rc : 0;
			if (rid < 0) { l.seq = -1; l.pos = 0; }
			lines[w] = l; line_req[w] = g; ++w; ++which;
		}
	}
}
#endif
