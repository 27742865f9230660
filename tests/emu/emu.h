/*
 * tests/emu/emu.h -- TEST INFRASTRUCTURE: a minimal host emulation of the HIP execution model
 * (grid of workgroups, 64-lane wavefronts, LDS, wave shuffles/ballots, __syncthreads, atomics).
 *
 * There is no GPU in the build container, so the product's kernel sources (speedseq_amd/csrc/k_*.h)
 * are additionally compiled against this header into tests/emu/libssgpu_emu.so, which lets the
 * `-m "not gpu"` tests run the *same kernel code* lane-for-lane on the CPU and compare it with the
 * oracle.  It is never loaded by the product (libssgpu.so fails loudly without a GPU).
 *
 * Model: workgroups run sequentially per host worker thread; every HIP thread is a fiber (own
 * stack, cooperative switch); a wave-level primitive is a rendezvous of the live lanes of the wave.
 */
#ifndef SSG_EMU_H
#define SSG_EMU_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <functional>
#include <algorithm>

namespace emu {
struct uint3_t { unsigned x, y, z; };
struct Fiber;
extern thread_local Fiber *cur;
extern thread_local uint3_t cur_tid, cur_bid, cur_bdim, cur_gdim;
extern thread_local char *dyn_lds;

void launch(unsigned grid, unsigned block, size_t lds_bytes, const std::function<void()> &body);
void yield();
/* wave rendezvous: publish v, wait for all live lanes, then read any lane's value through out[] */
void wave_exchange(uint64_t v, uint64_t out[64], uint64_t *live_mask);
void block_barrier();
}

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

#define threadIdx (emu::cur_tid)
#define blockIdx  (emu::cur_bid)
#define blockDim  (emu::cur_bdim)
#define gridDim   (emu::cur_gdim)

static inline void __syncthreads() { emu::block_barrier(); }

static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }

template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > v && !__atomic_compare_exchange_n(p, &o, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
template <class T> static inline T atomicCAS(T *p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

/* wave primitives used by the kernels (ssg_dev.h maps the wv_* wrappers onto these) */
static inline int emu_shfl_i32(int v, int src)
{
	uint64_t o[64], live; emu::wave_exchange((uint64_t)(uint32_t)v, o, &live);
	return (int)(uint32_t)o[src & 63];
}
static inline unsigned long long emu_ballot(int pred)
{
	uint64_t o[64], live, m = 0; emu::wave_exchange(pred ? 1 : 0, o, &live);
	for (int i = 0; i < 64; ++i) if ((live >> i & 1) && o[i]) m |= 1ull << i;
	return m;
}
#endif
