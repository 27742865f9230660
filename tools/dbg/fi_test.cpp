#include <stdio.h>
#include <stdlib.h>
#include <fcntl.h>
#include <zlib.h>
#include <chrono>
#include <algorithm>
#include "../../speedseq_amd/host/fast_inflate.h"
/* fi_test -- bin/bwa's gzip decoder (speedseq_amd/host/fast_inflate.h) on a file: tests compare size and CRC-32 of its output with zlib's.
 * usage: fi_test file.gz [crc] : decodes with fast_gz_t, prints bytes, crc32 of the output and the rate; exit 1 on error */
int main(int argc, char **argv)
{
	const int fd = open(argv[1], O_RDONLY); if (fd < 0) { perror("open"); return 2; }
	const bool do_crc = argc > 2;
	fast_gz_t g(fd);
	const auto t0 = std::chrono::steady_clock::now();
	unsigned long total = 0; uLong crc_all = crc32(0, 0, 0), crc_m = crc32(0, 0, 0); int members = 0;
	for (;;) {
		const uint8_t *p; bool mend;
		const long n = g.read_chunk(&p, (size_t)4 << 20, &mend);
		if (n < 0) { fprintf(stderr, "error: %s after %lu bytes (ip %zu iend %zu ireal %zu eof %d bitcnt %d st %d)\n", g.err ? g.err : "?", total, g.ip, g.iend, g.ireal, (int)g.eof_in, g.bitcnt, (int)g.st); return 1; }
		if (n) { total += (unsigned long)n; if (do_crc) { crc_all = crc32(crc_all, p, (uInt)n); crc_m = crc32(crc_m, p, (uInt)n); } }
		if (mend) { ++members; if (do_crc && (uint32_t)crc_m != g.crc_expect) { fprintf(stderr, "error: crc mismatch in member %d\n", members); return 1; } crc_m = crc32(0, 0, 0); }
		if (!n && !mend) break;
	}
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("%lu bytes, %d members, crc %08lx, %.3f s = %.1f MB/s out\n", total, members, (unsigned long)crc_all, dt, total / dt / 1e6);
	return 0;
}
