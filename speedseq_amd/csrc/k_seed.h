/*
 * k_seed.h -- gfx950 kernels for FM-index seeding.
 *
 *   (the product's seeding kernels are in k_smem2.h / ssg_seed.cpp: ssg_k_smem2 + ssg_k_smem_heavy)
 *   ssg_k_smem_lane one lane per read, nested loops as upstream writes them: the reference form on the device (SSG_SMEM_KERNEL=lane) and
 *                   the fall-back for reads whose lists outgrow the product kernels' capacities: the three SMEM passes of upstream
 *                   mem_collect_intv (bwt_smem1a x2 + bwt_seed_strategy1), intervals sorted by (start,end).
 *   ssg_k_smem_sort upstream's ks_introsort(mem_intv) per read, for the kernels that append in discovery order.
 *   ssg_k_sal_count per interval: number of sampled occurrences (<= max_occ) -> prefix sum.
 *   ssg_k_sal       one lane per (interval, occurrence): upstream bwt_sa LF-walk + sampled-SA
 *                   gather, then bns_intv2rid; writes mem_seed_t in upstream visiting order.
 */
#ifndef SSG_K_SEED_H
#define SSG_K_SEED_H
#include "ssg_dev.h"

/* interval vector; `st` = element stride: the per-lane work vectors are interleaved across the 64
 * lanes of a wave (entry e of lane l at base[e*64 + l]) so that lanes pushing / reading their e-th
 * entry together touch one contiguous 2-KB span instead of 64 scattered half-used lines */
struct ssg_ivec_t { ssg_intv_t *a; int n, cap; int ovf; unsigned nx; /* bwt_extend calls (rank-query accounting) */ int st; };
#define IV(v, i) ((v).a[(long)(i) * (v).st])
SSG_DEVFN void iv_push(ssg_ivec_t &v, const ssg_intv_t &x) { if (v.n < v.cap) IV(v, v.n) = x; else v.ovf = 1; ++v.n; }
SSG_DEVFN void iv_reverse(ssg_ivec_t &v)
{
	int n = v.n < v.cap ? v.n : v.cap;
	for (int j = 0; j < n >> 1; ++j) { ssg_intv_t t = IV(v, n-1-j); IV(v, n-1-j) = IV(v, j); IV(v, j) = t; }
}
SSG_DEVFN void ssg_set_intv(const ssg_index_view_t &ix, int c, ssg_intv_t &ik)
{
	ik.x0 = ix.L2[c] + 1; ik.x2 = ix.L2[c+1] - ix.L2[c]; ik.x1 = ix.L2[3-c] + 1; ik.info = 0;
}

/* upstream bwt_smem1a (max_intv == 0 as in mem_collect_intv) */
SSG_DEVFN int ssg_smem1(const ssg_index_view_t &ix, int len, const uint8_t *q, int x, uint64_t min_intv,
                        ssg_ivec_t &mem, ssg_ivec_t &va, ssg_ivec_t &vb)
{
	int i, j, c, ret;
	ssg_intv_t ik;
	ssg_ivec_t *prev = &va, *curr = &vb, *swap;
	mem.n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	ssg_set_intv(ix, q[x], ik);
	ik.info = (uint64_t)(x + 1);
	for (i = x + 1, curr->n = 0; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			const ssg_intv_t okc = ssg_bwt_extend1_lean(ix, ik, c, 0); ++mem.nx;
			if (okc.x2 != ik.x2) {
				iv_push(*curr, ik);
				if (okc.x2 < min_intv) break;
			}
			ik = okc; ik.info = (uint64_t)(i + 1);
		} else { iv_push(*curr, ik); break; }
	}
	if (i == len) iv_push(*curr, ik);
	iv_reverse(*curr);
	ret = (int)IV(*curr, 0).info;
	swap = curr; curr = prev; prev = swap;
	for (i = x - 1; i >= -1; --i) {
		c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		for (j = 0, curr->n = 0; j < prev->n; ++j) {
			ssg_intv_t p = IV(*prev, j);
			ssg_intv_t okc; okc.x0 = okc.x1 = okc.x2 = okc.info = 0;
			if (c >= 0) { okc = ssg_bwt_extend1_lean(ix, p, c, 1); ++mem.nx; }
			if (c < 0 || okc.x2 < min_intv) {
				if (curr->n == 0) {
					if (mem.n == 0 || (uint64_t)(i + 1) < (IV(mem, mem.n-1).info >> 32)) {
						ik = p; ik.info |= (uint64_t)(i + 1) << 32;
						iv_push(mem, ik);
					}
				}
			} else if (curr->n == 0 || okc.x2 != IV(*curr, curr->n-1).x2) {
				okc.info = p.info;
				iv_push(*curr, okc);
			}
		}
		if (curr->n == 0) break;
		swap = curr; curr = prev; prev = swap;
	}
	iv_reverse(mem);
	return ret;
}

/* upstream bwt_seed_strategy1 */
SSG_DEVFN int ssg_seed_strategy1(const ssg_index_view_t &ix, int len, const uint8_t *q, int x, int min_len, uint64_t max_intv, ssg_intv_t &mem, unsigned &nx)
{
	int i, c;
	ssg_intv_t ik;
	mem.x0 = mem.x1 = mem.x2 = mem.info = 0;
	if (q[x] > 3) return x + 1;
	ssg_set_intv(ix, q[x], ik);
	for (i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			const ssg_intv_t okc = ssg_bwt_extend1_lean(ix, ik, c, 0); ++nx;
			if (okc.x2 < max_intv && i - x >= min_len) {
				mem = okc;
				mem.info = (uint64_t)x << 32 | (uint64_t)(i + 1);
				return i + 1;
			}
			ik = okc;
		} else return i + 1;
	}
	return len;
}

struct ssg_intv_lt { SSG_DEVMEM bool operator()(const ssg_intv_t &a, const ssg_intv_t &b) const { return a.info < b.info; } };

/*
 * One lane per read.  seq: concatenated nt4 codes; off[r]..off[r+1] delimit read r.
 * out_intv: [n_reads x cap] ; out_n: per-read interval count (count > cap => overflow, the host
 * re-runs those reads with a larger cap).  scratch: per launched lane 3*scap intervals.
 */
__global__ void __launch_bounds__(64) ssg_k_smem_lane(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const int32_t *read_ids,
                           const uint8_t *seq, const int64_t *off,
                           ssg_intv_t *out_intv, int32_t *out_n, int cap,
                           ssg_intv_t *scratch, int scap, unsigned long long *n_extend)
{
	long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
	ssg_intv_t *my = scratch + (gt >> 6) * 3 * scap * 64 + (gt & 63);   /* wave slab, lane-interleaved */
	unsigned long long my_nx = 0;
	for (long it = gt; it < n_reads; it += nt) {
		int r = read_ids ? read_ids[it] : (int)it;
		const uint8_t *q = seq + off[r];
		int len = (int)(off[r+1] - off[r]);
		ssg_ivec_t mem = { out_intv + (long)it * cap, 0, cap, 0, 0, 1 };
		ssg_ivec_t mem1 = { my, 0, scap, 0, 0, 64 }, va = { my + (long)scap * 64, 0, scap, 0, 0, 64 }, vb = { my + 2L * scap * 64, 0, scap, 0, 0, 64 };
		int x = 0, i, k, old_n;
		int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
		if (len >= opt.min_seed_len) {
			while (x < len) { /* pass 1 */
				if (q[x] < 4) {
					x = ssg_smem1(ix, len, q, x, 1, mem1, va, vb);
					for (i = 0; i < mem1.n; ++i) {
						ssg_intv_t p = IV(mem1, i);
						int slen = (int)((uint32_t)p.info - (uint32_t)(p.info >> 32));
						if (slen >= opt.min_seed_len) iv_push(mem, p);
					}
				} else ++x;
			}
			old_n = mem.n < mem.cap ? mem.n : mem.cap; /* pass 2 */
			for (k = 0; k < old_n; ++k) {
				ssg_intv_t p = mem.a[k];
				int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
				if (end - start < split_len || p.x2 > (uint64_t)opt.split_width) continue;
				ssg_smem1(ix, len, q, (start + end) >> 1, p.x2 + 1, mem1, va, vb);
				for (i = 0; i < mem1.n; ++i) {
					ssg_intv_t m = IV(mem1, i);
					if ((int)((uint32_t)m.info - (uint32_t)(m.info >> 32)) >= opt.min_seed_len) iv_push(mem, m);
				}
			}
			if (opt.max_mem_intv > 0) { /* pass 3 */
				x = 0;
				while (x < len) {
					if (q[x] < 4) {
						ssg_intv_t m;
						x = ssg_seed_strategy1(ix, len, q, x, opt.min_seed_len, opt.max_mem_intv, m, mem1.nx);
						if (m.x2 > 0) iv_push(mem, m);
					} else ++x;
				}
			}
			if (mem.n <= mem.cap && !mem1.ovf && !va.ovf && !vb.ovf) ssg_introsort(mem.a, (long)mem.n, ssg_intv_lt());
		}
		out_n[it] = (mem.ovf || mem1.ovf || va.ovf || vb.ovf) ? -1 : mem.n;
		my_nx += mem1.nx;
	}
	if (n_extend && my_nx) atomicAdd(n_extend, my_nx);
}

/* one lane per read: intervals by (start,end), upstream's ks_introsort(mem_intv) */
__global__ void __launch_bounds__(64) ssg_k_smem_sort(int n_reads, ssg_intv_t *intv, const int32_t *n_intv, int cap)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const int n = n_intv[r];
	if (n > 1 && n <= cap) ssg_introsort(intv + r * cap, (long)n, ssg_intv_lt());
}

/* The same for the lists of up to LC intervals (nearly all) without introsort's per-lane control flow: the lanes of a wave sort lists of different lengths and
 * contents, so the wave walks the union of 64 different introsort paths, every compare a dependent load on the lane's own lines (3.7 ms per million pairs at
 * 1 % of the VALU rate, profiles/r06_pmc_sq.json).  Here a lane copies its records into LDS (four at a time, independent loads), holds the keys in registers,
 * ranks every key against the others in fully unrolled, predicated loops (no memory in the rank phase: a first form that read the keys from LDS in an n x n loop
 * took 4.5 ms, r06U), and writes record i to place rank(i).  LDS word (8 bytes) w of lane l at [w * 64 + l]: bank = lane whatever w.
 * The key is (start, end) = `info`; two intervals of a read with equal keys are intervals of the same pattern, hence the same 32 bytes (the order in which a
 * pattern was extended to does not change its interval: the premise of the table of short-pattern intervals as well), so every order of equal keys --
 * introsort's, or the stable one of the ranks here -- leaves the same list. */
template <int LC>
__global__ void __launch_bounds__(64) ssg_k_smem_sort_rank(int n_reads, ssg_intv_t *intv, const int32_t *n_intv, int cap, int32_t *todo, unsigned int *n_todo /* the reads with longer lists: ssg_k_smem_sort_wave */)
{
	__shared__ uint64_t L[LC * 3 * 64];   /* x0, x1, x2 of every record; the keys stay in registers */
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const int n = n_intv[r];
	if (n <= 1 || n > cap) return;
	ssg_intv_t *const p = intv + r * cap;
	if (n > LC) { todo[atomicAdd(n_todo, 1u)] = (int32_t)r; return; }   /* (one long list sorted by its lane held the whole wave: one read in twenty has one, nearly every wave had one) */
	uint64_t *const Ll = L + (threadIdx.x & 63);
	uint64_t key[LC];
	SSG_UNROLL for (int i0 = 0; i0 < LC; i0 += 4) {
		ssg_intv_t v[4];
		SSG_UNROLL for (int u = 0; u < 4; ++u) if (i0 + u < n) v[u] = p[i0 + u]; else { v[u].x0 = v[u].x1 = v[u].x2 = 0; v[u].info = ~0ull; }
		SSG_UNROLL for (int u = 0; u < 4; ++u) {
			key[i0 + u] = v[u].info;
			if (i0 + u < n) { Ll[((i0 + u) * 3 + 0) * 64] = v[u].x0; Ll[((i0 + u) * 3 + 1) * 64] = v[u].x1; Ll[((i0 + u) * 3 + 2) * 64] = v[u].x2; }
		}
	}
	SSG_UNROLL for (int i = 0; i < LC; ++i) {
		if (i < n) {
			int rank = 0;
			SSG_UNROLL for (int j = 0; j < LC; ++j) rank += (int)((j < n) & ((key[j] < key[i]) | ((key[j] == key[i]) & (j < i))));
			if (rank != i) { ssg_intv_t v; v.x0 = Ll[(i * 3 + 0) * 64]; v.x1 = Ll[(i * 3 + 1) * 64]; v.x2 = Ll[(i * 3 + 2) * 64]; v.info = key[i]; p[rank] = v; }
		}
	}
}

/* the longer lists, one WAVE per read: lane l holds records l, l + 64, ... (up to four: 256 intervals; beyond that lane 0 sorts), the keys lie in LDS and are
 * handed round by v_readlane as in k_sdp.h wv_rank_u64; rank = keys below + equal keys before (equal keys: identical records, see above). */
__global__ void __launch_bounds__(64) ssg_k_smem_sort_wave(ssg_intv_t *intv, const int32_t *n_intv, int cap, const int32_t *todo, const unsigned int *n_todo)
{
	__shared__ uint64_t sk[256];
	const int lane = wv_lane();
	const unsigned int nt = *n_todo;
	for (unsigned int t = blockIdx.x; t < nt; t += gridDim.x) {
		const long r = todo[t];
		const int n = n_intv[r];
		ssg_intv_t *const p = intv + r * cap;
		if (n > 256) { SSG_LANE0(ssg_introsort(p, (long)n, ssg_intv_lt())); continue; }
		ssg_intv_t rec[4];
		ssg_wave_ldssync();
		SSG_UNROLL for (int c = 0; c < 4; ++c) {
			const int me = c * 64 + lane;
			if (me < n) { rec[c] = p[me]; sk[me] = rec[c].info; } else { rec[c].x0 = rec[c].x1 = rec[c].x2 = 0; rec[c].info = ~0ull; }
		}
		ssg_wave_ldssync();
		int rk[4] = { 0, 0, 0, 0 };
		for (int j0 = 0; j0 < n; j0 += 64) {
			const unsigned long long kl = j0 + lane < n ? sk[j0 + lane] : ~0ull;
			SSG_UNROLL for (int u = 0; u < 64; ++u) {
				const unsigned long long k = (unsigned long long)wv_get64((long long)kl, u);
				SSG_UNROLL for (int c = 0; c < 4; ++c) rk[c] += (int)((k < rec[c].info) | ((k == rec[c].info) & (j0 + u < c * 64 + lane)));
			}
		}
		SSG_UNROLL for (int c = 0; c < 4; ++c) if (c * 64 + lane < n && rk[c] != c * 64 + lane) p[rk[c]] = rec[c];
	}
}

/* number of sampled occurrences of one interval (upstream mem_chain: step/count rule) */
SSG_DEVFN int ssg_intv_nocc(const ssg_mem_opt_t &opt, uint64_t x2)
{
	uint64_t step = x2 > (uint64_t)opt.max_occ ? x2 / (uint64_t)opt.max_occ : 1;
	uint64_t cnt = (x2 + step - 1) / step;
	return (int)(cnt < (uint64_t)opt.max_occ ? cnt : (uint64_t)opt.max_occ);
}

/* per read: total #occurrences over its intervals (for the prefix sum that places seeds) */
__global__ void ssg_k_sal_count(ssg_mem_opt_t opt, int n_reads, const ssg_intv_t *intv, const int32_t *n_intv, int cap, int32_t *n_seed, int32_t *pre /* optional [n_reads x cap]: occurrences of the read's earlier intervals */)
{
	long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	int n = n_intv[r], tot = 0;
	const ssg_intv_t *p = intv + r * cap;
	for (int i = 0; i < n; ++i) { if (pre) pre[r * cap + i] = tot; tot += ssg_intv_nocc(opt, p[i].x2); }
	n_seed[r] = tot;
}

/* read_of[first seed of read r] = r for the reads that have seeds (the array zeroed before): its running maximum is every seed's read */
__global__ void ssg_k_sal_mark(int n_reads, const int64_t *seed_off, int32_t *read_of)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < n_reads && seed_off[r + 1] > seed_off[r]) read_of[seed_off[r]] = (int32_t)r;
}

/*
 * One lane per (read, interval): walks that interval's sampled occurrences.  Seeds are written
 * at seed_off[r] + (occurrences of earlier intervals) + k, i.e. in upstream's visiting order.
 * Invalid seeds (bns_intv2rid < 0) get len = -1 and are skipped by the chaining kernel.
 */
__global__ void ssg_k_sal(ssg_index_view_t ix, ssg_mem_opt_t opt, int n_reads, const ssg_intv_t *intv, const int32_t *n_intv, int cap,
                          const int64_t *seed_off, ssg_seed_t *seeds, int32_t *seed_rid, const int32_t *pre /* from ssg_k_sal_count, or null */,
                          const int32_t *read_of /* per seed: its read (ssg_k_sal_mark + a running maximum), or null: bisection of seed_off */)
{	/* one lane per SEED (sampled occurrence): every lane does one independent <=31-step LF walk, so the
	 * random 64-byte fetches of a wave are 64 independent chains and long intervals cost no tail */
	const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= seed_off[n_reads]) return;
	long lo = 0, hi = n_reads;                       /* last r with seed_off[r] <= g */
	if (read_of) lo = read_of[g];                    /* (21 dependent loads for two million reads, by every lane) */
	else while (lo < hi) { long mid = (lo + hi + 1) >> 1; if (seed_off[mid] <= g) lo = mid; else hi = mid - 1; }
	const long r = lo;
	const ssg_intv_t *p = intv + r * cap;
	long k = g - seed_off[r];
	int ii = 0, c;
	if (pre) {   /* the seed's interval by bisection of the read's running counts: the walk along the list below is a dependent load per interval, and the 64 seeds
	              * of a wave that lies inside a repeat-heavy read (lists of a hundred intervals, hundreds of occurrences each) all walk most of it */
		const int32_t *pr = pre + r * cap;
		int lo2 = 0, hi2 = n_intv[r] - 1;
		while (lo2 < hi2) { const int mid = (lo2 + hi2 + 1) >> 1; if ((long)pr[mid] <= k) lo2 = mid; else hi2 = mid - 1; }
		ii = lo2; k -= pr[ii];
	} else
	while ((c = ssg_intv_nocc(opt, p[ii].x2)) <= k) { k -= c; ++ii; }
	const ssg_intv_t v = p[ii];
	const uint64_t step = v.x2 > (uint64_t)opt.max_occ ? v.x2 / (uint64_t)opt.max_occ : 1;
	ssg_seed_t s;
	s.rbeg = (int64_t)ssg_bwt_sa(ix, v.x0 + (uint64_t)k * step);
	s.qbeg = (int32_t)(v.info >> 32);
	s.len = s.score = (int)((uint32_t)v.info - (uint32_t)(v.info >> 32));
	s.next = -1;
	seeds[g] = s;
	seed_rid[g] = ssg_intv2rid_1(ix, s.rbeg, s.rbeg + s.len);
}
/* HBM copy of the suffix array sampled every `new_intv` rows instead of the file's sa_intv (upstream's .sa keeps every 32nd row; a seed
 * located through a denser table walks ~new_intv LF steps instead of ~32).  Each sampled row of the file starts a walk: LF(row with
 * SA = v) is the row with SA = v - 1, so the walk fills every row it passes whose index is a multiple of new_intv and ends at the next
 * sampled row -- the walks partition the LF cycle, every row is visited once (seq_len line fetches in all; asking bwt_sa for each new
 * sample instead walks ~sa_intv steps per sample: 7 x the fetches at 32 -> 4).  Walk lengths are geometric, so lanes take new
 * walks from a counter as they finish (one atomic per wave and refill). */
__global__ void ssg_k_sa_densify_walk(ssg_index_view_t ix, int new_intv, uint64_t *sa_new, unsigned long long n_old, unsigned long long *next, int refill_min)
{
	const uint64_t omask = (uint64_t)ix.sa_intv - 1, nmask = (uint64_t)new_intv - 1;
	int nshift = 0; while ((1 << nshift) < new_intv) ++nshift;
	const int lane = wv_lane();
	bool have = false, done = false; uint64_t r = 0, v = 0;
	for (;;) {
		/* idle lanes take new walks when refill_min of them wait (or nobody works): a refill is an atomic and two dependent loads that every working lane of the wave
		 * sits out -- taken every round (as until round 6) the walk made 5.9 G LF steps a second, a tenth of what this memory serves */
		const unsigned long long need = wv_ballot(!have && !done);
		if (need && (__popcll(need) >= refill_min || !wv_ballot(have))) {
			const int leader = __ffsll(need) - 1;
			unsigned long long base = 0;
			if (lane == leader) base = atomicAdd(next, (unsigned long long)__popcll(need));
			base = (unsigned long long)wv_bcast64((long long)base, leader);
			if (!have && !done) {
				const unsigned long long s = base + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
				if (s >= n_old) done = true;
				else { const uint64_t sv = ix.sa[s]; r = s * (uint64_t)ix.sa_intv; v = s == 0 ? ix.seq_len : sv; sa_new[r >> nshift] = sv; have = true; }
			}
		}
		if (!wv_ballot(have)) break;
		if (have) {
			if (r == ix.primary) r = 0;
			else r = ssg_lf_step(ix, r);
			--v;
			if ((r & omask) == 0) have = false;
			else if ((r & nmask) == 0) sa_new[r >> nshift] = v;
		}
	}
}
#endif
