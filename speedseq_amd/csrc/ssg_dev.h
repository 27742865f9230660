/*
 * ssg_dev.h -- device-side helpers for the gfx950 kernels: 64-lane wavefront primitives
 * (shuffles / ballots lower to DPP, ds_bpermute and s_ballot on CDNA4), FM-index rank on one
 * 64-byte block, 2-bit reference fetch.  Written for wave64 only.
 *
 * When SSG_EMU is defined the same sources build against tests/emu/emu.h (host emulation used
 * only by the CPU-side tests; see that header).
 */
#ifndef SSG_DEV_H
#define SSG_DEV_H
#include "ssg_types.h"

#ifdef SSG_EMU
#include "emu.h"
#define SSG_DEVFN static inline
#define SSG_DEVMEM inline
#define SSG_DEVFN_COLD static
SSG_DEVFN int wv_shfl(int v, int src) { return emu_shfl_i32(v, src); }
SSG_DEVFN unsigned long long wv_ballot(int p) { return emu_ballot(p); }
#define SSG_UNROLL
#else
#include <hip/hip_runtime.h>
#define SSG_DEVFN static __device__ __forceinline__
#define SSG_DEVMEM __device__ __forceinline__
#define SSG_DEVFN_COLD static __device__ __attribute__((noinline))   /* rarely taken paths: keep them out of the callers' register budget */
SSG_DEVFN int wv_shfl(int v, int src) { return __shfl(v, src, 64); }
SSG_DEVFN unsigned long long wv_ballot(int p) { return __ballot(p); }
#define SSG_UNROLL _Pragma("unroll")
#endif

#define SSG_WAVE 64
/* phase cycle counters for kernel tuning (read back with ssg_dbg_cycles) */
#ifdef SSG_EMU
static unsigned long long ssg_dbg_cyc[96];
#define SSG_TUNING 0
SSG_DEVFN unsigned long long ssg_clock() { return 0; }
#else
__device__ unsigned long long ssg_dbg_cyc[96];
#ifdef SSG_TUNE   /* `make lib TUNE=1`: instrumented build for tools/dbg/phase.py; the counters cost ~16 VGPRs in the SW kernels */
#define SSG_TUNING 1
SSG_DEVFN unsigned long long ssg_clock() { return (unsigned long long)clock64(); }
#else
#define SSG_TUNING 0
SSG_DEVFN unsigned long long ssg_clock() { return 0; }
#endif
#endif
/* make one lane's global stores visible to the other lanes of the same wave */
#ifdef SSG_EMU
SSG_DEVFN void ssg_wave_memsync() { (void)emu_ballot(0); } /* fibers are not lock-step: rendezvous */
#else
SSG_DEVFN void ssg_wave_memsync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
#endif
/* order one lane's LDS stores before the other lanes' LDS loads: the DS queue of a wave is in order, so on
 * the GPU only the compiler has to be held back; the emulator's fibers need a rendezvous */
#ifdef SSG_EMU
SSG_DEVFN void ssg_wave_ldssync() { (void)emu_ballot(0); }
#else
SSG_DEVFN void ssg_wave_ldssync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
#endif
/* A scalar (wave-uniform) mutation of memory shared by the wave: executed by lane 0 only, fenced on
 * both sides so every lane's earlier reads are done and every lane's later reads see it. */
#define SSG_LANE0(...) do { ssg_wave_memsync(); if (wv_lane() == 0) { __VA_ARGS__; } ssg_wave_memsync(); } while (0)

SSG_DEVFN int wv_lane() { return (int)(threadIdx.x & 63); }

/*
 * Cross-lane primitives.  On gfx950 they are DPP modifiers on ordinary VALU instructions
 * (row_shr:1/2/4/8, row_bcast:15/31, wave_shr:1 -- a few cycles each) and v_readlane_b32 (result in
 * an SGPR, i.e. broadcast for free); the generic __shfl lowers to ds_bpermute_b32, a round trip
 * through the LDS crossbar of >100 cycles, and a DP row needs ~40 of these on its critical path.
 */
#ifdef SSG_EMU
SSG_DEVFN int wv_prev(int v, int fill) { int l = wv_lane(); int r = wv_shfl(v, l - 1); return l == 0 ? fill : r; }
SSG_DEVFN int wv_get(int v, int src) { return wv_shfl(v, src); }          /* src wave-uniform */
SSG_DEVFN int wv_scan_max(int v)
{
	int l = wv_lane();
	for (int d = 1; d < 64; d <<= 1) { int o = wv_shfl(v, l - d); if (l >= d) v = v > o ? v : o; }
	return v;
}
SSG_DEVFN int wv_sum(int v) { for (int d = 1; d < 64; d <<= 1) v += wv_shfl(v, wv_lane() ^ d); return v; }
SSG_DEVFN int wv_scan_add(int v)
{
	int l = wv_lane();
	for (int d = 1; d < 64; d <<= 1) { int o = wv_shfl(v, l - d); if (l >= d) v += o; }
	return v;
}
#else
#define SSG_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)
SSG_DEVFN int wv_prev(int v, int fill) { return SSG_DPP(fill, v, 0x138 /* wave_shr:1 */, 0xf); }
SSG_DEVFN int wv_get(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
SSG_DEVFN int wv_scan_max(int v)
{	/* inclusive prefix max over lanes 0..lane: 4 row_shr steps inside each row of 16, then two row broadcasts */
	/* `old' = the identity of max: lets the compiler fold each step into one v_max_i32_dpp (with old = v it emits mov + mov_dpp + max) */
	int t;
	const int id = (int)0x80000000;
	t = SSG_DPP(id, v, 0x111, 0xf); v = v > t ? v : t;
	t = SSG_DPP(id, v, 0x112, 0xf); v = v > t ? v : t;
	t = SSG_DPP(id, v, 0x114, 0xf); v = v > t ? v : t;
	t = SSG_DPP(id, v, 0x118, 0xf); v = v > t ? v : t;
	t = SSG_DPP(id, v, 0x142 /* row_bcast:15 */, 0xa); v = v > t ? v : t;
	t = SSG_DPP(id, v, 0x143 /* row_bcast:31 */, 0xc); v = v > t ? v : t;
	return v;
}
SSG_DEVFN int wv_sum(int v)
{
	v += SSG_DPP(0, v, 0x111, 0xf); v += SSG_DPP(0, v, 0x112, 0xf); v += SSG_DPP(0, v, 0x114, 0xf); v += SSG_DPP(0, v, 0x118, 0xf);
	v += SSG_DPP(0, v, 0x142, 0xa); v += SSG_DPP(0, v, 0x143, 0xc);
	return __builtin_amdgcn_readlane(v, 63);
}
SSG_DEVFN int wv_scan_add(int v)
{	/* inclusive prefix sum over lanes 0..lane */
	v += SSG_DPP(0, v, 0x111, 0xf); v += SSG_DPP(0, v, 0x112, 0xf); v += SSG_DPP(0, v, 0x114, 0xf); v += SSG_DPP(0, v, 0x118, 0xf);
	v += SSG_DPP(0, v, 0x142, 0xa); v += SSG_DPP(0, v, 0x143, 0xc);
	return v;
}
#endif
SSG_DEVFN int wv_last(int v) { return wv_get(v, 63); }
/* dynamic work distribution: lane 0 takes the next index of a global queue, the wave shares it */
SSG_DEVFN long wv_queue_pop(unsigned int *queue)
{
	int k = 0;
	if (wv_lane() == 0) k = (int)atomicAdd(queue, 1u);
	return (long)(unsigned)wv_get(k, 0);
}
SSG_DEVFN int wv_bcast(int v, int src) { return wv_get(v, src); }
SSG_DEVFN long long wv_bcast64(long long v, int src)
{
	int lo = wv_shfl((int)(unsigned)(unsigned long long)v, src), hi = wv_shfl((int)((unsigned long long)v >> 32), src);
	return (long long)((unsigned long long)(unsigned)lo | (unsigned long long)(unsigned)hi << 32);
}
SSG_DEVFN long long wv_shfl64_xor(long long v, int d)
{
	const int src = wv_lane() ^ d;
	int lo = wv_shfl((int)(unsigned)(unsigned long long)v, src), hi = wv_shfl((int)((unsigned long long)v >> 32), src);
	return (long long)((unsigned long long)(unsigned)lo | (unsigned long long)(unsigned)hi << 32);
}
SSG_DEVFN long long wv_get64(long long v, int src)
{	/* src wave-uniform: two v_readlane */
	int lo = wv_get((int)(unsigned)(unsigned long long)v, src), hi = wv_get((int)((unsigned long long)v >> 32), src);
	return (long long)((unsigned long long)(unsigned)lo | (unsigned long long)(unsigned)hi << 32);
}
/* number of lanes below this one whose predicate holds, given the ballot */
SSG_DEVFN int wv_rank_of(unsigned long long ballot) { return __popcll(ballot & ((1ull << wv_lane()) - 1ull)); }
SSG_DEVFN int wv_max(int v) { return wv_last(wv_scan_max(v)); }
SSG_DEVFN int wv_min(int v) { return -wv_max(-v); }   /* |v| < 2^31 everywhere it is used */
SSG_DEVFN int imax(int a, int b) { return a > b ? a : b; }
SSG_DEVFN int imin(int a, int b) { return a < b ? a : b; }
SSG_DEVFN int iabs(int a) { return a < 0 ? -a : a; }
SSG_DEVFN long long lmax(long long a, long long b) { return a > b ? a : b; }
SSG_DEVFN long long lmin(long long a, long long b) { return a < b ? a : b; }

/* ---------------- FM-index rank: one 64-byte block per query ---------------- */
SSG_DEVFN int ssg_cnt16(uint32_t w, int c, int nsym)
{	/* # of 2-bit symbols == c among the first nsym (MSB-first) symbols of w */
	uint32_t m = ~(w ^ ((uint32_t)c * 0x55555555u));
	uint32_t t = m & (m >> 1) & 0x55555555u;
	if (nsym < 16) t &= nsym ? ~((1u << ((16 - nsym) << 1)) - 1) : 0u;
	return __popc(t);
}
/* occ of all four symbols in stored BWT [0..k] (k = with-$ row index, (uint64_t)-1 allowed) */
SSG_DEVFN void ssg_occ4(const ssg_index_view_t &ix, uint64_t k, uint64_t cnt[4])
{
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= ix.primary);
	const uint32_t *p = ix.bwt + ((k >> 7) << 4);
	struct alignas(16) q16 { uint32_t v[4]; };             /* the 64-byte block as four 16-byte loads */
	const q16 *pb = (const q16*)p;
	const q16 c0 = pb[0], c1 = pb[1], w0 = pb[2], w1 = pb[3];
	const uint32_t w[8] = { w0.v[0], w0.v[1], w0.v[2], w0.v[3], w1.v[0], w1.v[1], w1.v[2], w1.v[3] };
	const uint64_t pc[4] = { (uint64_t)c0.v[0] | (uint64_t)c0.v[1] << 32, (uint64_t)c0.v[2] | (uint64_t)c0.v[3] << 32,
	                         (uint64_t)c1.v[0] | (uint64_t)c1.v[1] << 32, (uint64_t)c1.v[2] | (uint64_t)c1.v[3] << 32 };
	const int r = (int)(k & 127) + 1;
	/* mask away the symbols past position r once per word, then count each base on the masked words */
	uint32_t mk[8];
	SSG_UNROLL for (int i = 0; i < 8; ++i) { int ns = r - i * 16; ns = ns < 0 ? 0 : ns > 16 ? 16 : ns; mk[i] = ns == 16 ? 0x55555555u : ns ? (~((1u << ((16 - ns) << 1)) - 1)) & 0x55555555u : 0u; }
	SSG_UNROLL for (int c = 0; c < 4; ++c) {
		int n = 0;
		SSG_UNROLL for (int i = 0; i < 8; ++i) { const uint32_t m = ~(w[i] ^ ((uint32_t)c * 0x55555555u)); n += __popc(m & (m >> 1) & mk[i]); }
		cnt[c] = pc[c] + (uint64_t)n;
	}
}
SSG_DEVFN uint64_t ssg_occ1(const ssg_index_view_t &ix, uint64_t k, int c)
{	/* upstream bwt_occ */
	if (k == ix.seq_len) return ix.L2[c+1] - ix.L2[c];
	if (k == (uint64_t)-1) return 0;
	k -= (k >= ix.primary);
	const uint32_t *p = ix.bwt + ((k >> 7) << 4);
	uint64_t n = ((const uint64_t*)p)[c];
	int r = (int)(k & 127) + 1;
	SSG_UNROLL for (int i = 0; i < 8; ++i) { int ns = r - i * 16; ns = ns < 0 ? 0 : ns > 16 ? 16 : ns; if (ns) n += ssg_cnt16(p[8 + i], c, ns); }
	return n;
}
SSG_DEVFN int ssg_bwt_sym(const ssg_index_view_t &ix, uint64_t k)
{	/* symbol at stored index k */
	const uint32_t *p = ix.bwt + ((k >> 7) << 4) + 8;
	return p[(k & 127) >> 4] >> ((~k & 15) << 1) & 3;
}
/* One LF step of upstream bwt_sa / bwt_invPsi: for a row k other than the primary, the row of the suffix one position to the left -- L2[c] + occ(k, c) with
 * c the row's BWT symbol.  Symbol and rank lie in the same 64-byte block (stored index k - (k > primary) for both), fetched here as four independent 16-byte
 * loads: one round trip and four requests a step, where ssg_bwt_sym + ssg_occ1 made up to ten 4-byte requests in two dependent trips -- and the request rate
 * of random lines, not their bytes, is what the locate stage and the denser-table walk run at. */
SSG_DEVFN uint64_t ssg_lf_step(const ssg_index_view_t &ix, uint64_t k)
{
	const uint64_t x = k - (k > ix.primary);
	struct alignas(16) q16 { uint32_t v[4]; };
	const q16 *pb = (const q16*)(ix.bwt + ((x >> 7) << 4));
	const q16 c01 = pb[0], c23 = pb[1], w0 = pb[2], w1 = pb[3];
	const uint32_t w[8] = { w0.v[0], w0.v[1], w0.v[2], w0.v[3], w1.v[0], w1.v[1], w1.v[2], w1.v[3] };
	const int r = (int)(x & 127), wi = r >> 4;
	uint32_t ww = 0;
	SSG_UNROLL for (int i = 0; i < 8; ++i) if (i == wi) ww = w[i];
	const int c = (int)(ww >> ((~r & 15) << 1)) & 3;
	const uint64_t cnt[4] = { (uint64_t)c01.v[0] | (uint64_t)c01.v[1] << 32, (uint64_t)c01.v[2] | (uint64_t)c01.v[3] << 32, (uint64_t)c23.v[0] | (uint64_t)c23.v[1] << 32, (uint64_t)c23.v[2] | (uint64_t)c23.v[3] << 32 };
	uint64_t n = 0;
	SSG_UNROLL for (int i = 0; i < 4; ++i) if (i == c) n = cnt[i];
	const int upto = r + 1;
	SSG_UNROLL for (int i = 0; i < 8; ++i) { int ns = upto - i * 16; ns = ns < 0 ? 0 : ns > 16 ? 16 : ns; n += (uint64_t)ssg_cnt16(w[i], c, ns); }
	return ix.L2[c] + n;
}
/* upstream bwt_extend */
SSG_DEVFN void ssg_bwt_extend(const ssg_index_view_t &ix, const ssg_intv_t &ik, ssg_intv_t ok[4], int is_back)
{
	uint64_t tk[4], tl[4];
	uint64_t kx = is_back ? ik.x0 : ik.x1, ox = is_back ? ik.x1 : ik.x0;
	ssg_occ4(ix, kx - 1, tk);
	ssg_occ4(ix, kx - 1 + ik.x2, tl);
	uint64_t nk[4], ns[4], no[4];
	for (int i = 0; i < 4; ++i) { nk[i] = ix.L2[i] + 1 + tk[i]; ns[i] = tl[i] - tk[i]; }
	no[3] = ox + (kx <= ix.primary && kx + ik.x2 - 1 >= ix.primary);
	no[2] = no[3] + ns[3];
	no[1] = no[2] + ns[2];
	no[0] = no[1] + ns[1];
	for (int i = 0; i < 4; ++i) {
		ok[i].x2 = ns[i];
		if (is_back) { ok[i].x0 = nk[i]; ok[i].x1 = no[i]; } else { ok[i].x1 = nk[i]; ok[i].x0 = no[i]; }
	}
}
/* upstream bwt_extend restricted to the one base the caller follows (all SMEM call sites use a
 * single ok[c]): same two rank blocks, a quarter of the live registers */
SSG_DEVFN ssg_intv_t ssg_bwt_extend1(const ssg_index_view_t &ix, const ssg_intv_t &ik, int c, int is_back)
{
	uint64_t tk[4], tl[4];
	const uint64_t kx = is_back ? ik.x0 : ik.x1, ox = is_back ? ik.x1 : ik.x0;
	ssg_occ4(ix, kx - 1, tk);
	ssg_occ4(ix, kx - 1 + ik.x2, tl);
	uint64_t no = ox + (kx <= ix.primary && kx + ik.x2 - 1 >= ix.primary);
	SSG_UNROLL for (int b = 3; b >= 0; --b) if (b > c) no += tl[b] - tk[b];
	uint64_t tkc = tk[0], tlc = tl[0];
	SSG_UNROLL for (int b = 1; b < 4; ++b) if (b == c) { tkc = tk[b]; tlc = tl[b]; }
	ssg_intv_t o;
	const uint64_t nk = ix.L2[c] + 1 + tkc;
	o.x2 = tlc - tkc; o.info = 0;
	if (is_back) { o.x0 = nk; o.x1 = no; } else { o.x1 = nk; o.x0 = no; }
	return o;
}
/* ssg_bwt_extend1 that fetches only the quarters of each rank block the followed base needs.  With T = k + 1 symbols up to and
 * including stored position k, sum over b > c of occ_b equals T - occ_0 (c = 0), T - occ_0 - occ_1 (c = 1), occ_3 (c = 2), 0 (c = 3):
 * at most two of the four running counts -- one 16-byte quarter -- and two popcount passes instead of four; of the symbol words
 * only those below the position: the third quarter, and the fourth only past the block's middle.
 * Both rank queries of an extension are issued together (one memory round trip), and when they fall into the same block -- the
 * rule once the interval is narrower than 128 rows, i.e. for most of a read -- the block is fetched once (upstream bwt_2occ4 has the
 * same special case for its cache).  On a multi-GB table the per-lane fetch is bound by the number of distinct lines and of load
 * instructions in flight, not by bytes. */
struct alignas(16) ssg_q16_t { uint32_t v[4]; };
/* among the first r (1..128) symbols of the block's 8 words: na = #symbols == (c & 2), nb = #symbols == (c & 2) + 1.
 * `hs': even bits set where the symbol's high bit matches, shifted so that only the first r symbols remain (a word wholly above r
 * is shifted by 31: bit 31 of the even-bit mask is never set). */
SSG_DEVFN void ssg_cnt_pair(const ssg_q16_t &w0, const ssg_q16_t &w1, int r, int c, int &na, int &nb)
{
	const uint32_t w[8] = { w0.v[0], w0.v[1], w0.v[2], w0.v[3], w1.v[0], w1.v[1], w1.v[2], w1.v[3] };
	const uint32_t hc = (c & 2) ? 0u : 0xffffffffu;
	const int r2 = r << 1;
	int nt = 0, n1 = 0;
	SSG_UNROLL for (int i = 0; i < 8; ++i) {
		int sh = 32 * (i + 1) - r2; sh = sh < 0 ? 0 : sh > 31 ? 31 : sh;
		const uint32_t hs = (((w[i] >> 1) ^ hc) & 0x55555555u) >> sh;
		nt += __popc(hs); n1 += __popc(hs & (w[i] >> sh));
	}
	na = nt - n1; nb = n1;
}
SSG_DEVFN ssg_intv_t ssg_bwt_extend1_lean(const ssg_index_view_t &ix, const ssg_intv_t &ik, int c, int is_back)
{
	const uint64_t kx = is_back ? ik.x0 : ik.x1, ox = is_back ? ik.x1 : ik.x0;
	const uint64_t k = kx - 1, l = k + ik.x2;
	const bool kneg = k == (uint64_t)-1, lneg = l == (uint64_t)-1;
	const uint64_t k2 = kneg ? 0 : k - (k >= ix.primary), l2 = lneg ? 0 : l - (l >= ix.primary);   /* k2 <= l2 */
	const ssg_q16_t *const pk = (const ssg_q16_t*)(ix.bwt + ((k2 >> 7) << 4)), *const pl = (const ssg_q16_t*)(ix.bwt + ((l2 >> 7) << 4));
	const int rk = (int)(k2 & 127) + 1, rl = (int)(l2 & 127) + 1;
	const bool same = (k2 >> 7) == (l2 >> 7);
	ssg_q16_t z; z.v[0] = z.v[1] = z.v[2] = z.v[3] = 0;
	/* all loads first: the upper query's block, then (other block only) the lower query's */
	const ssg_q16_t cql = pl[c >> 1], w0l = pl[2];
	ssg_q16_t w1l = z;
	if (rl > 64) w1l = pl[3];
	ssg_q16_t cqk = z, w0k = z, w1k = z;
	if (!same) { cqk = pk[c >> 1]; w0k = pk[2]; if (rk > 64) w1k = pk[3]; }
	/* same block: rk <= rl, so the fourth quarter is there whenever rk needs it (selected after the loads are out: a register copy
	 * of the upper block here would wait for it before the lower block's loads are issued) */
	uint32_t sm = same ? 0xffffffffu : 0u;            /* (x_l & sm) | x_k with x_k = 0 where nothing was loaded; opaque to the optimizer */
#if !defined(SSG_EMU) && !defined(SSG_NO_ASM_PINS)   /* SSG_NO_ASM_PINS: diagnostic builds only (tools/dbg/smem_variants.sh) */
	asm("" : "+v"(sm));
#endif
	SSG_UNROLL for (int i = 0; i < 4; ++i) { cqk.v[i] |= cql.v[i] & sm; w0k.v[i] |= w0l.v[i] & sm; w1k.v[i] |= w1l.v[i] & sm; }
	int nak, nbk, nal, nbl;
	ssg_cnt_pair(w0k, w1k, rk, c, nak, nbk);
	ssg_cnt_pair(w0l, w1l, rl, c, nal, nbl);
	const uint64_t oak = ((uint64_t)cqk.v[0] | (uint64_t)cqk.v[1] << 32) + (uint64_t)nak, obk = ((uint64_t)cqk.v[2] | (uint64_t)cqk.v[3] << 32) + (uint64_t)nbk;
	const uint64_t oal = ((uint64_t)cql.v[0] | (uint64_t)cql.v[1] << 32) + (uint64_t)nal, obl = ((uint64_t)cql.v[2] | (uint64_t)cql.v[3] << 32) + (uint64_t)nbl;
	const uint64_t Tk = k2 + 1, Tl = l2 + 1;
	uint64_t tkc = (c & 1) ? obk : oak, tkg = c == 0 ? Tk - oak : c == 1 ? Tk - oak - obk : c == 2 ? obk : 0;
	uint64_t tlc = (c & 1) ? obl : oal, tlg = c == 0 ? Tl - oal : c == 1 ? Tl - oal - obl : c == 2 ? obl : 0;
	if (kneg) tkc = tkg = 0;
	if (lneg) tlc = tlg = 0;
	const uint64_t no = ox + (kx <= ix.primary && kx + ik.x2 - 1 >= ix.primary) + (tlg - tkg);
	/* L2[c] by selects over scalars (left to itself the compiler makes it a per-lane load from the kernel-argument segment: a
	 * dependent memory round trip per extension) */
	uint64_t L0 = ix.L2[0], L1 = ix.L2[1], L2v = ix.L2[2], L3 = ix.L2[3];
#if !defined(SSG_EMU) && !defined(SSG_NO_ASM_PINS)
	asm("" : "+s"(L0)); asm("" : "+s"(L1)); asm("" : "+s"(L2v)); asm("" : "+s"(L3));
#endif
	const uint64_t l2c = c == 0 ? L0 : c == 1 ? L1 : c == 2 ? L2v : L3;
	ssg_intv_t o;
	const uint64_t nk = l2c + 1 + tkc;
	o.x2 = tlc - tkc; o.info = 0;
	if (is_back) { o.x0 = nk; o.x1 = no; } else { o.x1 = nk; o.x0 = no; }
	return o;
}

/* upstream bwt_sa: LF-walk to a sampled row */
SSG_DEVFN uint64_t ssg_bwt_sa(const ssg_index_view_t &ix, uint64_t k)
{
	uint64_t sa = 0, mask = (uint64_t)ix.sa_intv - 1;
	while (k & mask) {
		++sa;
		if (k == ix.primary) { k = 0; continue; }
		k = ssg_lf_step(ix, k);
	}
	return sa + ix.sa[k / (uint64_t)ix.sa_intv];
}

/* ---------------- reference access ---------------- */
SSG_DEVFN int ssg_ref_base(const ssg_index_view_t &ix, int64_t p)
{	/* base at doubled coordinate p in [0, 2*l_pac) */
	int64_t q = p < ix.l_pac ? p : (ix.l_pac << 1) - 1 - p;
	int b = ix.pac[q >> 2] >> ((~q & 3) << 1) & 3;
	return p < ix.l_pac ? b : 3 - b;
}
SSG_DEVFN int ssg_pos2rid(const ssg_index_view_t &ix, int64_t pos_f)
{	/* upstream bns_pos2rid */
	if (pos_f >= ix.l_pac) return -1;
	int left = 0, mid = 0, right = ix.n_ctg;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= ix.ctg_off[mid]) {
			if (mid == ix.n_ctg - 1) break;
			if (pos_f < ix.ctg_off[mid + 1]) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
SSG_DEVFN int64_t ssg_depos(const ssg_index_view_t &ix, int64_t pos, int *is_rev)
{
	return (*is_rev = (pos >= ix.l_pac)) ? (ix.l_pac << 1) - 1 - pos : pos;
}
SSG_DEVFN int ssg_intv2rid(const ssg_index_view_t &ix, int64_t rb, int64_t re)
{	/* upstream bns_intv2rid */
	int is_rev;
	if (rb < ix.l_pac && re > ix.l_pac) return -2;
	int rid_b = ssg_pos2rid(ix, ssg_depos(ix, rb, &is_rev));
	int rid_e = rb < re ? ssg_pos2rid(ix, ssg_depos(ix, re - 1, &is_rev)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}
/* the same with one bisection: the contig of the first base, then whether the last base lies in it too (bns_pos2rid of a position below l_pac is the contig that
 * holds it, so rid_e == rid_b exactly when it does) */
SSG_DEVFN int ssg_intv2rid_1(const ssg_index_view_t &ix, int64_t rb, int64_t re)
{
	int is_rev;
	if (rb < ix.l_pac && re > ix.l_pac) return -2;
	const int rid_b = ssg_pos2rid(ix, ssg_depos(ix, rb, &is_rev));
	if (rb >= re || rid_b < 0) return rid_b;
	const int64_t pe = ssg_depos(ix, re - 1, &is_rev);
	const int64_t lo = ix.ctg_off[rid_b], hi = rid_b == ix.n_ctg - 1 ? ix.l_pac : ix.ctg_off[rid_b + 1];
	return pe >= lo && pe < hi ? rid_b : -1;
}
SSG_DEVFN int ssg_score(const ssg_mem_opt_t &o, int t, int q) { return o.mat[t * 5 + q]; }

/* ---------------- klib-compatible introsort over an index permutation ----------------
 * Same operation sequence as ks_introsort (htslib ksort.h:178-229), so ties land in the same
 * order as the reference's unstable sort.  `LT(a,b)` compares two element *values*. */
template <class T, class LT>
SSG_DEVFN void ssg_insertsort(T *s, T *t, LT lt)
{
	for (T *i = s + 1; i < t; ++i)
		for (T *j = i; j > s && lt(*j, *(j - 1)); --j) { T x = *j; *j = *(j - 1); *(j - 1) = x; }
}
template <class T, class LT>
SSG_DEVFN void ssg_combsort(T *a, long n, LT lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	int do_swap; long gap = n;
	do {
		if (gap > 2) { gap = (long)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		do_swap = 0;
		for (T *i = a; i < a + n - gap; ++i) {
			T *j = i + gap;
			if (lt(*j, *i)) { T x = *i; *i = *j; *j = x; do_swap = 1; }
		}
	} while (do_swap || gap > 2);
	if (gap != 1) ssg_insertsort(a, a + n, lt);
}
template <class T, class LT>
SSG_DEVFN void ssg_introsort(T *a, long n, LT lt)
{
	struct { T *l, *r; int d; } stack[40]; int top = 0;   /* the larger side is pushed: depth <= log2(n) */
	int d; T rp, *s, *t, *i, *j, *k;
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) { T x = a[0]; a[0] = a[1]; a[1] = x; } return; }
	for (d = 2; (1l << d) < n; ++d);
	s = a; t = a + (n - 1); d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { ssg_combsort(s, t - s + 1, lt); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
			else k = lt(*j, *i) ? i : j;
			rp = *k;
			if (k != t) { T x = *k; *k = *t; *t = x; }
			for (;;) {
				do ++i; while (lt(*i, rp));
				do --j; while (i <= j && lt(rp, *j));
				if (j <= i) break;
				T x = *i; *i = *j; *j = x;
			}
			{ T x = *i; *i = *t; *t = x; }
			if (i - s > t - i) {
				if (i - s > 16) { stack[top].l = s; stack[top].r = i - 1; stack[top].d = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stack[top].l = i + 1; stack[top].r = t; stack[top].d = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { ssg_insertsort(a, a + n, lt); return; }
			--top; s = stack[top].l; t = stack[top].r; d = stack[top].d;
		}
	}
}
/* The same three sorts over an array VIEW (a.get(i) / a.set(i, x), int indices) instead of a pointer: for arrays that are not contiguous
 * (lane-interleaved LDS words, k_chain.h).  Operation for operation the pointer forms above; the stack of pending ranges is four 16-bit
 * entries in one register (a range is pushed only when it is the LARGER part of its parent and longer than 16: for n <= 256 there are
 * never more than four), so nothing of it lives in scratch memory. */
template <class A, class LT>
SSG_DEVFN void ssg_insertsort_ix(A a, int s, int t, LT lt)
{
	for (int i = s + 1; i < t; ++i)
		for (int j = i; j > s; --j) { const auto x = a.get(j), y = a.get(j - 1); if (!lt(x, y)) break; a.set(j, y); a.set(j - 1, x); }
}
template <class A, class LT>
SSG_DEVFN void ssg_combsort_ix(A a, int s, int n, LT lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	int do_swap; int gap = n;
	do {
		if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		do_swap = 0;
		for (int i = s; i < s + n - gap; ++i) {
			const auto x = a.get(i), y = a.get(i + gap);
			if (lt(y, x)) { a.set(i, y); a.set(i + gap, x); do_swap = 1; }
		}
	} while (do_swap || gap > 2);
	if (gap != 1) ssg_insertsort_ix(a, s, s + n, lt);
}
template <class A, class LT>
SSG_DEVFN void ssg_introsort_ix(A a, int n, LT lt)   /* n <= 256 */
{
	unsigned long long stack = 0; int top = 0;   /* pending ranges, newest in the low 16 bits: first index | last index << 8 */
	int d, s, t, i, j, k;
	if (n < 1) return;
	if (n == 2) { const auto x = a.get(0), y = a.get(1); if (lt(y, x)) { a.set(0, y); a.set(1, x); } return; }
	for (d = 2; (1 << d) < n; ++d);
	s = 0; t = n - 1; d <<= 1;
	unsigned int dstack = 0;   /* the depth counters of the pushed ranges, 8 bits each */
	for (;;) {
		if (s < t) {
			if (--d == 0) { ssg_combsort_ix(a, s, t - s + 1, lt); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			{
				const auto vk = a.get(k), vi = a.get(i), vj = a.get(j);
				if (lt(vk, vi)) { if (lt(vk, vj)) k = j; }
				else k = lt(vj, vi) ? i : j;
			}
			const auto rp = a.get(k);
			if (k != t) { const auto x = a.get(t); a.set(k, x); a.set(t, rp); }
			for (;;) {
				auto vi = rp, vj = rp;
				do { ++i; vi = a.get(i); } while (lt(vi, rp));
				do { --j; if (i <= j) vj = a.get(j); } while (i <= j && lt(rp, vj));
				if (j <= i) break;
				a.set(i, vj); a.set(j, vi);
			}
			{ const auto x = a.get(i), y = a.get(t); a.set(i, y); a.set(t, x); }
			if (i - s > t - i) {
				if (i - s > 16) { stack = stack << 16 | (unsigned long long)(s | (i - 1) << 8); dstack = dstack << 8 | (unsigned)d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stack = stack << 16 | (unsigned long long)((i + 1) | t << 8); dstack = dstack << 8 | (unsigned)d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { ssg_insertsort_ix(a, 0, n, lt); return; }
			--top; s = (int)(stack & 255); t = (int)(stack >> 8 & 255); d = (int)(dstack & 255); stack >>= 16; dstack >>= 8;
		}
	}
}
#endif
