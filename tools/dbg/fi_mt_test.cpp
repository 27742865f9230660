/* fi_mt_test -- the several-thread gzip decoder (speedseq_amd/host/fast_inflate_mt.h) on a file: size and CRC-32 of its output, rate.
 * usage: fi_mt_test file.gz threads [chunk_bytes] [nocrc] */
#include <stdio.h>
#include <stdlib.h>
#include <fcntl.h>
#include <zlib.h>
#include <chrono>
#include "../../speedseq_amd/host/fast_inflate_mt.h"
int main(int argc, char **argv)
{
	const int fd = open(argv[1], O_RDONLY); if (fd < 0) { perror("open"); return 2; }
	fast_gz_mt_t g(fd, argc > 2 ? atoi(argv[2]) : 4, argc > 3 ? (size_t)atol(argv[3]) : (size_t)2 << 20);
	const bool do_crc = argc <= 4;
	const auto t0 = std::chrono::steady_clock::now();
	unsigned long total = 0; uLong crc_all = crc32(0, 0, 0), crc_m = crc32(0, 0, 0); int members = 0; bool crc_bad = false;
	const bool ok = g.run([&](const uint8_t *p, size_t n, bool mend, uint32_t expect) {
		if (n) { total += n; if (do_crc) { crc_all = crc32(crc_all, p, (uInt)n); crc_m = crc32(crc_m, p, (uInt)n); } }
		if (mend) { ++members; if (do_crc && (uint32_t)crc_m != expect) crc_bad = true; crc_m = crc32(0, 0, 0); }
		return true;
	});
	if (!ok || crc_bad) { fprintf(stderr, "error: %s after %lu bytes\n", crc_bad ? "crc mismatch" : g.err ? g.err : "?", total); return 1; }
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("%lu bytes, %d members, crc %08lx, %.3f s = %.1f MB/s out\n", total, members, (unsigned long)crc_all, dt, total / dt / 1e6);
	return 0;
}
