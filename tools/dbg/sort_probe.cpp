// Does hipcub::DeviceRadixSort::SortKeys on a bit range of 64-bit keys depend on what its temporary storage held before?
// (round 5: the mate-rescue job list came back unsorted when the arena handed the sort a block with old contents)
// hipcc --offload-arch=gfx950 -O2 tools/dbg/sort_probe.cpp -o tools/dbg/sort_probe
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static int run(long n, int b0, int b1, int poison, int out_poison)
{
	std::vector<uint64_t> h((size_t)n), o((size_t)n);
	uint64_t x = 88172645463325252ull;
	for (long i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[(size_t)i] = (uint64_t)(160 + 8 * (x % 3)) << 48 | (uint64_t)(500 + (x >> 8) % 300) << 32 | (uint64_t)i; }
	uint64_t *din, *dout; CK(hipMalloc(&din, n * 8)); CK(hipMalloc(&dout, n * 8));
	CK(hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice));
	CK(hipMemset(dout, out_poison ? 0xA5 : 0, n * 8));
	size_t tb = 0; CK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, din, dout, (int)n, b0, b1));
	void *tmp; CK(hipMalloc(&tmp, tb + 256)); CK(hipMemset(tmp, poison ? 0xA5 : 0, tb + 256));
	CK(hipcub::DeviceRadixSort::SortKeys(tmp, tb, din, dout, (int)n, b0, b1, 0));
	CK(hipDeviceSynchronize());
	CK(hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost));
	const uint64_t mask = (b1 >= 64 ? ~0ull : (1ull << b1) - 1) & ~((1ull << b0) - 1);
	std::vector<uint64_t> ref(h); std::stable_sort(ref.begin(), ref.end(), [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
	long bad = 0, disorder = 0; for (long i = 0; i < n; ++i) bad += ref[(size_t)i] != o[(size_t)i];
	for (long i = 1; i < n; ++i) disorder += (o[(size_t)i] & mask) < (o[(size_t)i - 1] & mask);
	std::vector<uint64_t> a(h), b(o); std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
	printf("n %8ld bits [%d,%d) temp %zu bytes %s: %ld of %ld differ from a stable sort; %ld descents on the sorted bits; permutation of the input: %s\n", n, b0, b1, tb, poison ? "poisoned" : "zeroed", bad, n, disorder, a == b ? "yes" : "NO");
	CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(tmp));
	return bad != 0;
}
int main()
{
	int rc = 0;
	for (long n : { 24700L, 640000L, 3000L })
		for (int b1 : { 64, 58, 40 })
			for (int p = 0; p < 2; ++p) rc |= run(n, 32, b1, p, p);
	return rc;
}
