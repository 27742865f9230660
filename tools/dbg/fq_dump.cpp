/* fq_dump -- the records bin/bwa's FASTQ reader (speedseq_amd/host/fastq.h) sees in a file, one per line, and the reader's end state:
 * tests compare the several-thread reader of plain files with the one-thread reader on the same bytes.  `-r` only counts (ingest rate). */
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include "../../speedseq_amd/host/fastq.h"
int main(int argc, char **argv)
{
	bool rate = false; int ai = 1;
	if (ai < argc && !strcmp(argv[ai], "-r")) { rate = true; ++ai; }
	if (ai >= argc) { fprintf(stderr, "usage: fq_dump [-r] <reads.fq>\n"); return 2; }
	gzFile fp = gzopen(argv[ai], "r");
	if (!fp) { perror(argv[ai]); return 1; }
	const auto t0 = std::chrono::steady_clock::now();
	long n = 0, bases = 0; int rc;
	{
		fq_feed_t feed(fp, true, 16384, argv[ai]);
		fq_cursor_t c(feed);
		const fq_block_t *b; int i;
		while ((rc = c.next(&b, &i)) == 0) {
			++n; bases += b->seq_o[i + 1] - b->seq_o[i];
			if (rate) continue;
			fputs(b->txt.data() + b->name_o[i], stdout); putchar('\t');
			if (b->com_o[i] != UINT32_MAX) fputs(b->txt.data() + b->com_o[i], stdout);
			putchar('\t');
			for (uint32_t k = b->seq_o[i]; k < b->seq_o[i + 1]; ++k) putchar("ACGTN"[b->seq[k]]);
			putchar('\t');
			if (b->has_q[i]) fputs(b->qual.data() + b->qual_o[i], stdout); else putchar('*');
			putchar('\n');
		}
	}
	gzclose(fp);
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (rate) fprintf(stderr, "%ld records, %ld bases in %.3f s: %.2f M records/s\n", n, bases, dt, n / dt / 1e6);
	printf("end\t%d\t%ld\n", rc, n);
	return 0;
}
