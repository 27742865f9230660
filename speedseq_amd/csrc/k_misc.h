/*
 * k_misc.h -- small data-movement kernels between pipeline stages (HBM streaming, one lane per item).
 */
#ifndef SSG_K_MISC_H
#define SSG_K_MISC_H
#include "k_pair.h"

/* copy each read's region list into its (larger) slice of the pairing-stage array: a wave takes 64 reads, one after the other, 8 bytes a lane (a lane per read copied
 * the thousands of 88-byte regions of a repeat read alone: 2.0 ms of the step, all of it that tail) */
__global__ void ssg_k_copy_regs(int n_reads, const int64_t *src_off, const ssg_alnreg_t *src, const int32_t *n_reg, const int64_t *dst_off, ssg_alnreg_t *dst)
{
	static_assert(sizeof(ssg_alnreg_t) % 8 == 0, "ssg_alnreg_t is copied in 8-byte words");
	constexpr int W = (int)(sizeof(ssg_alnreg_t) / 8);
	const long r0 = (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 6;
	const int lane = wv_lane();
	const long rm = r0 + lane;
	const long so = rm < n_reads ? (long)src_off[rm] : 0, dof = rm < n_reads ? (long)dst_off[rm] : 0;
	const int nr = rm < n_reads ? n_reg[rm] : 0;
	for (int k = 0; k < 64 && r0 + k < n_reads; ++k) {
		const int n = wv_get(nr, k);
		if (!n) continue;
		const uint64_t *s = (const uint64_t*)(src + wv_get64(so, k)); uint64_t *d = (uint64_t*)(dst + wv_get64(dof, k));
		for (int t = lane; t < n * W; t += 64) d[t] = s[t];
	}
}

/* gather each read's requests into a dense array */
__global__ void ssg_k_compact_req(int n_reads, const int64_t *src_off, const ssg_alnreq_t *src, const int32_t *n_req, const int64_t *dst_off, ssg_alnreq_t *dst)
{
	long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const ssg_alnreq_t *s = src + src_off[r]; ssg_alnreq_t *d = dst + dst_off[r];
	for (int i = 0; i < n_req[r]; ++i) d[i] = s[i];
}

/* ---------------- duplicate marking (upstream samblaster, row a14) ---------------- */
/* primary record of one end as samblaster sees it */
typedef struct { uint64_t k0, k1, k2; } ssg_sig_t;

SSG_DEVFN uint64_t ssg_mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

/* one lane per pair: 5'-unclipped signature (oracle/orc_samblaster.c header) + 64-bit hash */
__global__ void ssg_k_sig(long n_pairs, const ssg_sbl_end_t *ends, ssg_sig_t *sig, uint64_t *hash, uint32_t *ord)
{
	long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	uint64_t k[2][3]; int m[2];
	for (int e = 0; e < 2; ++e) {
		const ssg_sbl_end_t x = ends[2*p + e];
		m[e] = !(x.flag & 0x4) && x.seq >= 0;
		uint64_t strand = (x.flag & 0x10) ? 1 : 0;
		int64_t q = strand ? (int64_t)x.pos + x.ralen - 1 + x.rclip : (int64_t)x.pos - x.lclip;
		k[e][0] = (uint64_t)(uint32_t)x.seq; k[e][1] = (uint64_t)(q + (1LL << 31)); k[e][2] = strand;
	}
	ssg_sig_t s;
	if (m[0] && m[1]) {
		int swap = k[0][0] > k[1][0] || (k[0][0] == k[1][0] && (k[0][1] > k[1][1] || (k[0][1] == k[1][1] && k[0][2] > k[1][2])));
		const uint64_t *lo = swap ? k[1] : k[0], *hi = swap ? k[0] : k[1];
		s.k0 = lo[0] << 32 | hi[0]; s.k1 = lo[1] << 1 | lo[2]; s.k2 = hi[1] << 1 | hi[2];
	} else if (m[0] || m[1]) {
		const uint64_t *a = m[0] ? k[0] : k[1];
		s.k0 = a[0]; s.k1 = a[1] << 1 | a[2]; s.k2 = 0;
	} else { s.k0 = s.k1 = s.k2 = ~0ull; } /* never a duplicate */
	sig[p] = s;
	hash[p] = (m[0] || m[1]) ? ssg_mix64(s.k0 ^ ssg_mix64(s.k1 ^ ssg_mix64(s.k2))) : ~0ull - (uint64_t)p;
	ord[p] = (uint32_t)p;
}

/* ---------------- small device-side bookkeeping (keeps counts, offsets and work orders off the host) ---------------- */
__global__ void ssg_k_iota(int32_t *a, long n) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = (int32_t)i; }
__global__ void ssg_k_scan_tail(const int32_t *in, int64_t *out, long n) { if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = n > 0 ? out[n-1] + in[n-1] : 0; }
/* cnt[0] = #(key > tC), cnt[1] = #(tB < key <= tC), cnt[2] = #(max(tA,1) <= key <= tB), cnt[3] = #(key < 0), cnt[4] = #(key != 0) */
__global__ void ssg_k_class_counts(const int32_t *key, long n, int tA, int tB, int tC, unsigned int *cnt)
{	/* one atomic per wave and class */
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const int k = i < n ? key[i] : 0;
	const unsigned long long b0 = wv_ballot(k > tC), b1 = wv_ballot(k <= tC && k > tB), b2 = wv_ballot(k <= tB && k >= tA && k > 0), b3 = wv_ballot(k < 0), b4 = wv_ballot(k != 0);
	if (wv_lane() == 0) {
		if (b0) atomicAdd(&cnt[0], (unsigned)__popcll(b0));
		if (b1) atomicAdd(&cnt[1], (unsigned)__popcll(b1));
		if (b2) atomicAdd(&cnt[2], (unsigned)__popcll(b2));
		if (b3) atomicAdd(&cnt[3], (unsigned)__popcll(b3));
		if (b4) atomicAdd(&cnt[4], (unsigned)__popcll(b4));
	}
}
/* cnt[j] = #(key > t[j]) for ten thresholds at once, given the order that sorts the keys descending: a binary search per threshold (a lane each) instead of a pass
 * over the keys with an atomic per wave and threshold (1.6 ms per step for 2 M reads, profiles/r05q_chain_stage_timeline_before.txt) */
struct ssg_thr6_t { int t[10]; };
__global__ void ssg_k_count_gt6(const int32_t *key, const int32_t *order, long n, ssg_thr6_t th, unsigned int *cnt)
{
	const int j = (int)threadIdx.x;
	if (j >= 10) return;
	const int t = th.t[j];
	long lo = 0, hi = n;   /* first place whose key is <= t */
	while (lo < hi) { const long mid = (lo + hi) >> 1; if (key[order[mid]] > t) lo = mid + 1; else hi = mid; }
	cnt[j] = (unsigned int)lo;
}
/* pairing-stage capacities per read: region slots (own regions + up to 4 rescued hits per anchor of the mate) and request slots */
__global__ void ssg_k_pair_caps(int n_reads, const int32_t *n_reg, int max_matesw, int32_t *cap2, int32_t *capq, int32_t *pair_key)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const int m = n_reg[r ^ 1], c = n_reg[r] + 4 * (m < max_matesw ? m : max_matesw) + 4;
	cap2[r] = c; capq[r] = 2 * c + 2;
	if (!(r & 1)) pair_key[r >> 1] = n_reg[r] + n_reg[r + 1];
}
#endif
