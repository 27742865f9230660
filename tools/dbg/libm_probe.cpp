// Device libm (ocml, double) vs host glibc on the arguments of upstream's decision path:
//   mem_approx_mapq_se: log(l) for integer l, log(seedcov), log(sub_n + 1)          (k_pair.h:351-358)
//   mem_pair:           .721 * log(2 * erfc(|ns| / sqrt 2)), ns = (dist - avg) / std (k_pair.h:397)
// Every value then feeds (int)(x + .499).  The probe evaluates each expression on both sides over the whole argument range
// the aligner can produce, reports the largest distance in ulps and how many of the (int)(x + .499) decisions would differ for
// ANY integer offset added before the truncation (i.e. whether a value crosses k + .501 between the two libraries).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dbg/libm_probe.cpp -o tools/dbg/libm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_log(const double *x, double *y, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = log(x[i]); }
__global__ void k_pair(const double *ns, double *y, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = .721 * log(2. * erfc(fabs(ns[i]) * M_SQRT1_2)); }

static long long ulps(double a, double b) { int64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8); if (x < 0) x = INT64_MIN - x; if (y < 0) y = INT64_MIN - y; return llabs((long long)(x - y)); }
static bool flips(double a, double b) { return floor(a + .499 + 1000.) != floor(b + .499 + 1000.); }   /* (int)(k + x + .499) for integer k >= -1000 */

static void compare(const char *what, const std::vector<double> &arg, const std::vector<double> &dev, const std::vector<double> &host, double scale)
{
	long long mx = 0; long nd = 0, nf = 0; size_t worst = 0;
	for (size_t i = 0; i < arg.size(); ++i) {
		if (std::isinf(host[i]) && std::isinf(dev[i])) continue;
		long long u = ulps(dev[i], host[i]); if (u) ++nd; if (u > mx) { mx = u; worst = i; }
		if (flips(dev[i] * scale, host[i] * scale)) ++nf;
	}
	printf("%-44s %9zu arguments  %8ld differ  max %lld ulp (at %.17g)  (int)(x+.499) decisions that differ: %ld\n", what, arg.size(), nd, mx, arg[worst], nf);
}

int main()
{
	{	/* log of every integer the MAPQ formulas can see (read / seed-coverage lengths, sub_n + 1) */
		std::vector<double> a; for (int l = 1; l <= 100000; ++l) a.push_back((double)l);
		std::vector<double> h(a.size()), d(a.size()); for (size_t i = 0; i < a.size(); ++i) h[i] = log(a[i]);
		double *dx, *dy; CK(hipMalloc(&dx, a.size() * 8)); CK(hipMalloc(&dy, a.size() * 8));
		CK(hipMemcpy(dx, a.data(), a.size() * 8, hipMemcpyHostToDevice));
		k_log<<<(a.size() + 255) / 256, 256>>>(dx, dy, (long)a.size()); CK(hipMemcpy(d.data(), dy, a.size() * 8, hipMemcpyDeviceToHost));
		compare("log(l), l = 1..100000", a, d, h, 1.0);
		compare("4.343 * log(n), n = 1..100000", a, d, h, 4.343);
		compare("30 * log(seedcov) (x 1 - sub/score <= 1)", a, d, h, 30.0);
	}
	{	/* the pairing term over insert-size models (avg, std) and every distance in the model's window */
		/* mem_pair only scores distances inside the model's window [low, high]: |ns| stays below ~5 (avg +- 4 std, or the
		 * quartile fence when wider); the tail up to |ns| = 38 (where erfc underflows to 0 and log gives -inf) is listed separately */
		std::vector<double> a, tail;
		srand48(7);
		for (int m = 0; m < 400; ++m) {
			const double avg = 150 + drand48() * 700, sd = 5 + drand48() * 150;
			for (int dist = 0; dist <= 2500; ++dist) { const double ns = (dist - avg) / sd; (fabs(ns) <= 8 ? a : tail).push_back(ns); }
		}
		for (int pass = 0; pass < 2; ++pass) {
		if (pass) a.swap(tail);
		std::vector<double> h(a.size()), d(a.size()); for (size_t i = 0; i < a.size(); ++i) h[i] = .721 * log(2. * erfc(fabs(a[i]) * M_SQRT1_2));
		double *dx, *dy; CK(hipMalloc(&dx, a.size() * 8)); CK(hipMalloc(&dy, a.size() * 8));
		CK(hipMemcpy(dx, a.data(), a.size() * 8, hipMemcpyHostToDevice));
		k_pair<<<(a.size() + 255) / 256, 256>>>(dx, dy, (long)a.size()); CK(hipMemcpy(d.data(), dy, a.size() * 8, hipMemcpyDeviceToHost));
		compare(pass ? ".721 log(2 erfc(|ns|/sqrt 2)), 8 < |ns| < 40" : ".721 log(2 erfc(|ns|/sqrt 2)), |ns| <= 8", a, d, h, 1.0);
		}
	}
	return 0;
}
