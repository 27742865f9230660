// test utility: the records of a plain FASTQ file as rank 0's scanners see them (ranksplit.h) -- the one-thread scanner, then the several-thread one --
// one line per record (start, end, sequence length), then the return code and the message; `-q` prints only a checksum and the rates
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <functional>
#include <memory>
#include <atomic>
#include <chrono>
#include <zlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>
#include "fastq.h"
#include "ranks.h"
#include "ranksplit.h"
template <class S> static void run(S &A, bool quiet, const char *what)
{
	const auto t0 = std::chrono::steady_clock::now();
	size_t r0 = 0, r1 = 0, len = 0, n = 0; uint64_t h = 1469598103934665603ull; int k;
	while ((k = A.next(&r0, &r1, &len)) == 1) { ++n; if (!quiet) printf("%zu %zu %zu\n", r0, r1, len); h = (h ^ r0) * 1099511628211ull; h = (h ^ r1) * 1099511628211ull; h = (h ^ len) * 1099511628211ull; }
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("%s: rc %d records %zu at %zu hash %016llx why [%s]\n", what, k, n, (size_t)A.at, (unsigned long long)h, A.why.c_str());
	if (quiet) fprintf(stderr, "%s: %.3f s, %.2f GB/s, %.2f M records/s\n", what, dt, (double)r1 / 1e9 / dt, n / 1e6 / dt);
}
int main(int argc, char **argv)
{
	const bool quiet = argc > 2 && !strcmp(argv[2], "-q");
	{ rs_scan_t A(rs_file_reader(argv[1])); run(A, quiet, "one thread"); }
	{ rs_pscan_t P(argv[1]); run(P, quiet, "several threads"); }
	return 0;
}
