// Bench / profile utility (not part of the product library): the synthetic GRCh37-shaped reference of SURVEY.md 8d in ONE kernel
// launch -- an order-k Markov chain over ACGT sampled by independent streams, each a contiguous stretch of the genome.
// (bench.py used ~35 k small torch dispatches for this; rocprofv3's counter collection does not survive that many.)
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/synth/synth_ref.cpp -o tools/synth/libsynthref.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t sm64(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

// cdf: 4^order x 3 thresholds (float); stream s writes out[s * per .. (s + 1) * per)
__global__ void k_markov(uint8_t *out, int64_t L, int64_t per, const float *cdf, int order, uint64_t seed)
{
	const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t b = s * per, e = b + per < L ? b + per : L;
	if (b >= L) return;
	const uint32_t mask = (1u << (2 * order)) - 1;
	uint64_t r = sm64(seed ^ sm64((uint64_t)s));
	uint32_t ctx = (uint32_t)(r >> 11) & mask;
	for (int64_t i = b; i < e; ++i) {
		r = sm64(r);
		const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
		const float *t = cdf + 3 * (size_t)ctx;
		const uint32_t nx = (u > t[0]) + (u > t[1]) + (u > t[2]);
		out[i] = (uint8_t)nx;
		ctx = ((ctx << 2) | nx) & mask;
	}
}

extern "C" int synth_markov(uint8_t *d_out, int64_t L, int64_t n_streams, const float *d_cdf, int order, uint64_t seed)
{
	const int64_t per = (L + n_streams - 1) / n_streams;
	hipLaunchKernelGGL(k_markov, dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, 0, d_out, L, per, d_cdf, order, seed);
	return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : -1;
}
