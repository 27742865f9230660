// int32 VALU issue probe for MI355X: the roofline denominator of the Smith-Waterman kernels (integer DP: no MFMA, a few bytes
// of HBM per thousand cell updates).  Independent v_add_u32 / v_max_i32 chains, enough of them per lane to cover the pipeline,
// at 4 / 8 / 16 waves per CU-SIMD set.  Reports lane-operations per second and the implied cycles per wave64 instruction.
// hipcc --offload-arch=gfx950 -O3 tools/dbg/valu_probe.cpp -o tools/dbg/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MIX>   // 0: add only, 1: add + max (the DP's mix), 2: packed 16-bit add + max
__global__ void __launch_bounds__(256) probe(int iters, int seed, int *sink)
{
	int a[8];
	for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x * (k + 1);
	const int b = seed | 1, c = seed ^ 0x5555;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			if (MIX == 0) { a[k] += b; a[k] += c; }
			else if (MIX == 1) { a[k] += b; a[k] = a[k] > c ? a[k] : c; }
			else {
				typedef short s2 __attribute__((ext_vector_type(2)));
				s2 x = __builtin_bit_cast(s2, a[k]) + __builtin_bit_cast(s2, b);
				x = __builtin_elementwise_max(x, __builtin_bit_cast(s2, c));
				a[k] = __builtin_bit_cast(int, x);
			}
		}
	}
	int s = 0;
	for (int k = 0; k < 8; ++k) s ^= a[k];
	if (s == 0x12345678) sink[0] = s;
}

template <int MIX> static void run(int nblocks, int iters, int *sink, const char *name)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	probe<MIX><<<nblocks, 256>>>(100, 3, sink); CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	probe<MIX><<<nblocks, 256>>>(iters, 3, sink);
	CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double wave_instr = (double)nblocks * 4 * iters * 16;      // 16 VALU instructions per iteration per wave
	const double lane_ops = wave_instr * 64;
	printf("%-28s waves/CU %5.1f  %8.2f ms  %7.2f T lane-ops/s  %5.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, nblocks * 4 / 256.0, ms,
	       lane_ops / ms / 1e9, (ms * 1e-3 * 2.4e9) / (wave_instr / 1024.0));
}

int main()
{
	int *sink; CK(hipMalloc(&sink, 64));
	for (int nb : { 256, 512, 1024, 2048 }) {
		run<0>(nb, 20000, sink, "v_add_u32 x2");
		run<1>(nb, 20000, sink, "v_add_u32 + v_max_i32");
		run<2>(nb, 20000, sink, "v_pk_add_u16 + v_pk_max_i16");
	}
	return 0;
}
