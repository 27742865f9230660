// Random 64-byte-line gather probe for MI355X: what rate of dependent random line reads does the memory
// system sustain, by access shape?  (Roofline denominator for the FM-index kernels: every bwt_extend is
// two random 64-byte rank blocks.)  hipcc --offload-arch=gfx950 -O3 tools/dbg/gather_probe.cpp -o /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

__global__ void fill(uint4 *buf, size_t n16) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint64_t a = mix(i), b = mix(a); buf[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)); } }

// mode 0: one 16-B load per step; 1: the four 16-B quarters of one line; 2: two lines x four quarters (bwt_extend shape);
// 3: quad-cooperative, one line per quad per step (each lane one quarter); 4: quad-cooperative, two lines per step
template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint4 *buf, uint64_t nlines, int iters, uint64_t *sink)
{
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t s = mix(MODE >= 3 ? gid >> 2 : gid), acc = 0;
	const int qlane = threadIdx.x & 3;
	for (int it = 0; it < iters; ++it) {
		const uint64_t l1 = s % nlines, l2 = mix(s) % nlines;
		uint32_t v = 0;
		if (MODE == 0) { uint4 a = buf[l1 * 4 + (s >> 40 & 3)]; v = a.x ^ a.w; }
		else if (MODE == 1) { for (int k = 0; k < 4; ++k) { uint4 a = buf[l1 * 4 + k]; v += __popc(a.x) + __popc(a.y) + __popc(a.z) + a.w; } }
		else if (MODE == 2) { for (int k = 0; k < 4; ++k) { uint4 a = buf[l1 * 4 + k], b = buf[l2 * 4 + k]; v += __popc(a.x) + __popc(a.y) + a.w + __popc(b.x) + __popc(b.z) + b.w; } }
		else if (MODE == 3) { uint4 a = buf[l1 * 4 + qlane]; v = __popc(a.x) + __popc(a.y) + a.w; v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); }
		else { uint4 a = buf[l1 * 4 + qlane], b = buf[l2 * 4 + qlane]; v = __popc(a.x) + a.w + __popc(b.y) + b.w; v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); }
		acc += v;
		s = mix(s ^ v);
	}
	if (acc == 0x1234567) sink[0] = acc;
}

// streaming twin for the counter calibration: every lane reads 16 B, consecutive lanes consecutive addresses, a known byte count per launch
// (the microarchitecture guide: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads; tools/pmc_summarize.py measures the unit)
__global__ void __launch_bounds__(256) stream_read(const uint4 *buf, size_t n16, uint64_t *sink)
{
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 a = buf[i]; acc += a.x ^ a.w; }
	if (acc == 0x1234567u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) stream_write(uint4 *buf, size_t n16)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) buf[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

template <int MODE> static void run(const uint4 *buf, uint64_t nlines, int nblocks, int iters, uint64_t *sink, const char *name, double gb)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	probe<MODE><<<nblocks, 256>>>(buf, nlines, 50, sink);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	probe<MODE><<<nblocks, 256>>>(buf, nlines, iters, sink);
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	const double lanes = (double)nblocks * 256, lines_per_step = MODE == 0 ? 1 : MODE == 1 ? 1 : MODE == 2 ? 2 : MODE == 3 ? 0.25 : 0.5;
	const double lines = lanes * iters * lines_per_step;
	printf("%-34s buf %5.2f GB  waves/CU %5.1f  %8.2f ms  %7.2f G lines/s  (%7.1f GB/s at 64 B)  %6.2f us per step\n", name, gb, nblocks * 4 / 256.0, ms, lines / ms / 1e6,
	       lines * 64 / ms / 1e6, ms * 1e3 / iters);
}

int main(int argc, char **argv)
{
	const double gbs[] = { 1.0, 3.1, 4.0 };   /* 3.1 GB: the rank blocks (64 B per 128 symbols of fwd + revcomp) of a 3.1 Gbp reference */
	uint64_t *sink; CK(hipMalloc(&sink, 8));
	for (double gb : gbs) {
		const size_t bytes = (size_t)(gb * (1ull << 30));
		uint4 *buf; CK(hipMalloc(&buf, bytes));
		fill<<<4096, 256>>>(buf, bytes / 16);
		CK(hipDeviceSynchronize());
		const uint64_t nlines = bytes / 64;
		for (int nb : { 1024, 2048 }) { // x4 waves per block / 256 CUs
			const int iters = 400;
			run<0>(buf, nlines, nb, iters, sink, "lane: one 16 B load", gb);
			run<1>(buf, nlines, nb, iters, sink, "lane: one line (4 x 16 B)", gb);
			run<2>(buf, nlines, nb, iters, sink, "lane: two lines (8 x 16 B)", gb);
			run<3>(buf, nlines, nb, iters, sink, "quad: one line (1 x 16 B per lane)", gb);
			run<4>(buf, nlines, nb, iters, sink, "quad: two lines", gb);
		}
		if (gb == 4.0) {   /* the streaming launches (names stream_read / stream_write in the counter CSV): 4 GiB read, 4 GiB written, past the 256 MB Infinity Cache */
			hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float ms;
			for (int rep = 0; rep < 2; ++rep) {
				CK(hipEventRecord(a)); stream_read<<<8192, 256>>>(buf, bytes / 16, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
				printf("stream_read  %zu bytes  %8.2f ms  %7.1f GB/s\n", bytes, ms, bytes / ms / 1e6);
			}
			for (int rep = 0; rep < 2; ++rep) {
				CK(hipEventRecord(a)); stream_write<<<8192, 256>>>(buf, bytes / 16); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
				printf("stream_write %zu bytes  %8.2f ms  %7.1f GB/s\n", bytes, ms, bytes / ms / 1e6);
			}
		}
		CK(hipFree(buf));
	}
	return 0;
}
