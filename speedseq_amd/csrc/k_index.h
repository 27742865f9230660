/*
 * k_index.h -- FM-index construction on the MI355X (upstream `bwa index`: bwtindex.c bwa_idx_build ->
 * bns_fasta2bntseq, bwt_pac2bwt, bwt_bwtupdate_core, bwt_cal_sa).
 *
 * Upstream builds the BWT of T = fwd || revcomp(fwd) with a sequential induced-sorting / BWT-merge
 * program; here the suffix array itself is built in HBM (288 GB hold the 8-byte SA and inverse SA of a
 * 6.2 G-symbol human T with room to spare) by a bucketed radix sort of 34-symbol prefixes followed by
 * prefix doubling restricted to the suffixes that are not yet unique, and the on-disk structures are
 * derived from it by streaming kernels.  All positions and ranks are 64-bit.
 *
 * Order convention (identical to upstream's BWT with an implicit smallest terminator): a suffix that
 * runs off the end of T is padded with 'A' (code 0) for the first-round key, and in every doubling
 * round a suffix whose second half starts past the end gets a second key below every real rank,
 * ordered so that the shorter suffix sorts first ("S$" < "SA$" < "SAA...").
 *
 * Every kernel is one lane per item (grid-stride: a HIP launch holds fewer than 2^32 threads, the text 6.2 G symbols), streaming HBM; the sorts / scans / compactions are hipCUB.
 */
#ifndef SSG_K_INDEX_H
#define SSG_K_INDEX_H
#include "ssg_dev.h"

#define SSG_IDX_KEYSYM 32          /* symbols in the first-round 64-bit key (after the bucket prefix) */

/* T[i] for i < n = 2*l_pac: forward strand then its reverse complement */
__global__ void ssg_k_idx_text(const uint8_t *fwd, int64_t l_pac, uint8_t *T, int64_t n_pad)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t n = 2 * l_pac;
		T[i] = i < l_pac ? (uint8_t)(fwd[i] & 3) : i < n ? (uint8_t)(3 - (fwd[n - 1 - i] & 3)) : (uint8_t)0;
	}
}

/* The suffixes of bucket b listed in text order, without a flag array in between (until round 5: flag kernel + a library reduction + a library
 * select per bucket, each over the whole text -- 128 passes of 74 ms at 6.2 G symbols, three quarters of the build).  A workgroup owns 4096
 * consecutive positions, a lane 16 of them (the list keeps the order of the text): _count leaves the bucket's suffixes per workgroup,
 * _scatter writes them behind the exclusive prefix sums of those counts. */
#define SSG_IDX_BK_PER_LANE 16
#define SSG_IDX_BK_PER_WG (256 * SSG_IDX_BK_PER_LANE)
SSG_DEVFN uint32_t ssg_idx_bucket_mask(const uint8_t *T, int64_t n, int p, uint32_t b, int64_t i0)
{	/* bit k: suffix i0 + k is in bucket b.  The lane's 16 symbols and the p - 1 <= 15 behind them as two 16-byte loads (i0 is a multiple of 16,
	 * T is zero-padded by >= 64 bytes past n: consecutive lanes read consecutive 16-byte words, the second load mostly hits the first of the next lane) */
	uint32_t m = 0, v = 0;
	if (i0 >= n) return 0;
	if (p == 0) { for (int k = 0; k < SSG_IDX_BK_PER_LANE; ++k) if (i0 + k < n) m |= 1u << k; return m; }   /* one bucket: every suffix */
	uint32_t x[8];
	memcpy(x, T + i0, 32);
	uint64_t w = 0;   /* the 32 symbols two bits each, the first on top */
	SSG_UNROLL for (int j = 0; j < 8; ++j) { const uint32_t t = x[j]; w = w << 8 | (uint64_t)((t & 3u) << 6 | (t >> 8 & 3u) << 4 | (t >> 16 & 3u) << 2 | (t >> 24 & 3u)); }
	const uint32_t keep = p >= 16 ? ~0u : (1u << (2 * p)) - 1u;
	SSG_UNROLL for (int k = 0; k < SSG_IDX_BK_PER_LANE; ++k) {   /* p + k <= 31: p <= 8 (SSG_INDEX_BUCKET_P), k <= 15 */
		v = (uint32_t)(w >> (2 * (32 - p - k))) & keep;
		if (i0 + k < n && v == b) m |= 1u << k;
	}
	return m;
}
__global__ void __launch_bounds__(256) ssg_k_idx_bucket_count(const uint8_t *T, int64_t n, int p, uint32_t b, uint32_t *wg_cnt)
{
	__shared__ uint32_t part[4];
	const int64_t i0 = (int64_t)blockIdx.x * SSG_IDX_BK_PER_WG + (int64_t)threadIdx.x * SSG_IDX_BK_PER_LANE;
	const int c = __popc(ssg_idx_bucket_mask(T, n, p, b, i0));
	const int w = wv_sum(c);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (uint32_t)w;
	__syncthreads();
	if (threadIdx.x == 0) wg_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ void __launch_bounds__(256) ssg_k_idx_bucket_scatter(const uint8_t *T, int64_t n, int p, uint32_t b, const uint64_t *wg_off, uint64_t *pos)
{
	__shared__ uint32_t part[4];
	const int64_t i0 = (int64_t)blockIdx.x * SSG_IDX_BK_PER_WG + (int64_t)threadIdx.x * SSG_IDX_BK_PER_LANE;
	uint32_t m = ssg_idx_bucket_mask(T, n, p, b, i0);
	const int c = __popc(m);
	const int incl = wv_scan_add(c);
	if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = (uint32_t)incl;
	__syncthreads();
	uint64_t at = wg_off[blockIdx.x] + (uint64_t)(incl - c);
	for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) at += part[w];
	while (m) { const int k = __ffsll((unsigned long long)m) - 1; m &= m - 1; pos[at++] = (uint64_t)(i0 + k); }
}

/* first-round key of the suffixes listed in pos[]: symbols p .. p+31, two bits each, first symbol most significant */
__global__ void ssg_k_idx_key(const uint8_t *T, const uint64_t *pos, int64_t m, int p, uint64_t *key)
{
	for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
		const uint8_t *s = T + pos[j] + p;
		uint64_t v = 0;
		SSG_UNROLL for (int k = 0; k < SSG_IDX_KEYSYM; ++k) v = v << 2 | s[k];
		key[j] = v;
	}
}

/* head[j] = j when element j starts a run of equal sorted keys, else 0 (inclusive max scan -> index of the run's head) */
__global__ void ssg_k_idx_heads(const uint64_t *ks, int64_t m, int64_t *head)
{
	for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
		head[j] = (j == 0 || ks[j] != ks[j - 1]) ? j : 0;
	}
}

/* one sorted bucket into the suffix array: SA[base + j] = position, inverse SA = index of the group head;
 * grp[j] = that index, pend[j] = 1 when the group has more than one member (goes to the doubling rounds) */
__global__ void ssg_k_idx_place(const uint64_t *ps, const int64_t *gh, int64_t m, uint64_t base, uint64_t *SA, uint64_t *rank, uint64_t *grp, uint8_t *pend)
{
	for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t pos = ps[j];
		const int64_t h = gh[j];
		SA[base + (uint64_t)j] = pos;
		rank[pos] = base + (uint64_t)h;
		grp[j] = base + (uint64_t)h;
		const bool single = h == j && (j + 1 == m || gh[j + 1] == j + 1);
		pend[j] = (uint8_t)!single;
	}
}

/* doubling round: second key of a pending suffix = rank of the suffix h symbols further on */
__global__ void ssg_k_idx_key2(const uint64_t *pos, int64_t m, uint64_t h, uint64_t n, const uint64_t *rank, uint64_t *key2, uint64_t *iota)
{
	for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t q = pos[e] + h;
		key2[e] = q < n ? rank[q] + n : n - 1 - pos[e];   /* past the end: below every real rank, shorter suffix first */
		iota[e] = (uint64_t)e;
	}
}
__global__ void ssg_k_idx_gather(const uint64_t *perm, const uint64_t *src, int64_t m, uint64_t *dst)
{
	for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) dst[e] = src[perm[e]];
}
/* after the (group, key2) sort: oh[e] = e at the first element of an old group, else 0 */
__global__ void ssg_k_idx_oldheads(const uint64_t *grp_s, int64_t m, int64_t *oh)
{
	for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
		oh[e] = (e == 0 || grp_s[e] != grp_s[e - 1]) ? e : 0;
	}
}
/* SA index of every pending element (its old group's first index + its place inside the group) and
 * nh[e] = that index where a new (group, key2) run starts, else 0 (SA indices grow with e, so a max scan finds the run head) */
__global__ void ssg_k_idx_newheads(const uint64_t *grp_s, const uint64_t *key2_s, const int64_t *ohs, int64_t m, uint64_t *saidx, int64_t *nh)
{
	for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t si = grp_s[e] + (uint64_t)(e - ohs[e]);
		saidx[e] = si;
		const bool head = e == 0 || grp_s[e] != grp_s[e - 1] || key2_s[e] != key2_s[e - 1];
		nh[e] = head ? (int64_t)si : 0;
	}
}
__global__ void ssg_k_idx_replace(const uint64_t *pos_s, const uint64_t *saidx, const int64_t *nhs, int64_t m, uint64_t *SA, uint64_t *rank, uint64_t *grp_new, uint8_t *pend)
{
	for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t pos = pos_s[e], si = saidx[e];
		SA[si] = pos;
		rank[pos] = (uint64_t)nhs[e];
		grp_new[e] = (uint64_t)nhs[e];
		const bool single = (uint64_t)nhs[e] == si && (e + 1 == m || (uint64_t)nhs[e + 1] == saidx[e + 1]);
		pend[e] = (uint8_t)!single;
	}
}

/* stored BWT symbol k (the with-$ matrix minus the primary row; upstream bwt_pac2bwt): row r = k + (k >= primary),
 * row 0 = "$" is preceded by T[n-1], row r >= 1 is the suffix SA[r-1] and is preceded by T[SA[r-1] - 1] */
__global__ void ssg_k_idx_bwt_sym(const uint8_t *T, const uint64_t *SA, uint64_t n, uint64_t primary, uint8_t *B)
{
	for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t r = k + (k >= primary);
		B[k] = r == 0 ? T[n - 1] : T[SA[r - 1] - 1];
	}
}
/* one 128-symbol block: 8 packed words (16 symbols each, first symbol in the top bits) + the block's symbol counts */
__global__ void ssg_k_idx_bwt_pack(const uint8_t *B, uint64_t n, uint64_t nblk, uint32_t *words /* nblk x 8 */, uint32_t *cnt /* 4 x nblk */)
{
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t c[4] = {0, 0, 0, 0};
		for (int w = 0; w < 8; ++w) {
			uint32_t v = 0;
			for (int s = 0; s < 16; ++s) {
				const uint64_t k = b * 128 + (uint64_t)w * 16 + (uint64_t)s;
				const uint32_t x = k < n ? B[k] : 0u;
				v = v << 2 | x;
				if (k < n) { c[0] += x == 0; c[1] += x == 1; c[2] += x == 2; c[3] += x == 3; }
			}
			words[b * 8 + (uint64_t)w] = v;
		}
		for (int i = 0; i < 4; ++i) cnt[(uint64_t)i * nblk + b] = c[i];
	}
}
/* the interleaved .bwt body (upstream bwt_bwtupdate_core): per block 4 x u64 running counts then its symbol words;
 * the final counts follow the last symbol word.  occ: 4 x (nblk + 1) exclusive prefix sums of cnt. */
__global__ void ssg_k_idx_bwt_write(const uint32_t *words, const uint64_t *occ, uint64_t n, uint64_t nblk, uint32_t *bwt)
{
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= nblk; b += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t n_words = (n + 15) / 16;
		uint32_t *o = b < nblk ? bwt + b * 16 : bwt + n_words + 8 * nblk;
		for (int i = 0; i < 4; ++i) { const uint64_t v = occ[(uint64_t)i * (nblk + 1) + b]; o[2 * i] = (uint32_t)v; o[2 * i + 1] = (uint32_t)(v >> 32); }
		if (b < nblk) {
			const uint64_t left = n_words - b * 8;                 /* symbol words of this block that exist in the file */
			for (uint64_t w = 0; w < 8 && w < left; ++w) o[8 + w] = words[b * 8 + w];
		}
	}
}
__global__ void ssg_k_idx_find_primary(const uint64_t *SA, uint64_t n, uint64_t *out)
{	/* with-$ row of suffix 0 */
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) if (SA[r] == 0) *out = r + 1;
}
/* sampled suffix array (upstream bwt_cal_sa): sample j = SA value of with-$ row j * intv; row 0 holds (uint64_t)-1 */
__global__ void ssg_k_idx_sa_sample(const uint64_t *SA, uint64_t n_sa, int intv, uint64_t *samp)
{
	for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_sa; j += (uint64_t)gridDim.x * blockDim.x) {
		samp[j] = j == 0 ? (uint64_t)-1 : SA[j * (uint64_t)intv - 1];
	}
}
/* .pac: four bases per byte, first base in the top bits (upstream _set_pac) */
__global__ void ssg_k_idx_pac(const uint8_t *fwd, int64_t l_pac, uint8_t *pac, int64_t nbytes)
{
	for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nbytes; b += (int64_t)gridDim.x * blockDim.x) {
		uint32_t v = 0;
		for (int k = 0; k < 4; ++k) { const int64_t i = b * 4 + k; v = v << 2 | (i < l_pac ? (uint32_t)(fwd[i] & 3) : 0u); }
		pac[b] = (uint8_t)v;
	}
}
__global__ void ssg_k_idx_stride_u64(const uint64_t *src, uint64_t n_out, int stride, uint64_t *dst)
{
	for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += (uint64_t)gridDim.x * blockDim.x) dst[j] = src[j * (uint64_t)stride];
}
#endif
