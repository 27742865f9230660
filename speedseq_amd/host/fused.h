/*
 * fused.h -- the binary hand-off between this repository's `bwa mem`, `samblaster` and `sambamba` (SURVEY.md 7.1 "fused mode").
 *
 * The reference wires its stages with SAM text on pipes (/root/reference/bin/speedseq:438-441); at MI355X alignment rates the text
 * (0.8 GB/s to print, parse, re-print and parse again) is the wall.  When speedseq.config -- which the script `source`s, so the
 * variable reaches every stage -- exports SSG_FUSED=1, `bwa mem` writes the records it would have printed as BAM records (the bytes
 * `sambamba view -S -f bam` makes of its lines) in frames, `samblaster` takes its decisions on those records (same device entry point,
 * same options, the line view derived from the BAM fields instead of the text), patches FLAG / MC / MQ and forwards them,
 * `sambamba view` passes the frames through and `sambamba sort` reads them directly.  The side streams stay SAM text (the script
 * pipes them through gawk): bwa attaches the text of the few pairs that can qualify (supplementary lines, or both ends mapped
 * and not a proper pair), samblaster emits from that text exactly as in text mode.  The text path remains the parity path; both must
 * produce the same three BAM files (tests/test_fused.py).  `bwa mem -C` (speedseq realign: another program sits between bwa and
 * samblaster there) never fuses.
 *
 * Stream: 8 magic bytes, then frames { u32 type, u32 0, u64 payload bytes }.
 *   HEADER  SAM header text (every stage may append its @PG line)
 *   BATCH   bwa -> samblaster: fu_batch_t, candidate table, candidate text, BAM records
 *   MAIN    samblaster -> sort: BAM records (block_size-prefixed, htslib sam.c:443-473)
 *   END     no payload
 */
#ifndef SSG_FUSED_H
#define SSG_FUSED_H
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <errno.h>
#include <unistd.h>

#define FU_MAGIC "SSGFUSE1"
enum { FU_HEADER = 1, FU_BATCH = 2, FU_MAIN = 3, FU_END = 4 };
struct fu_frame_t { uint32_t type, zero; uint64_t len; };
/* BATCH payload: fu_batch_t | fu_cand_t[n_cand] | text[text_bytes] | bam[bam_bytes].  A candidate names its first record (ordinal within
 * the batch) and owns text[text_off .. next candidate's text_off): the SAM lines of those records, one per record, in order. */
struct fu_batch_t { uint64_t n_rec, bam_bytes, n_cand, text_bytes; };
struct fu_cand_t { uint64_t first_rec, n_rec, text_off; };

static inline bool fu_enabled() { const char *e = getenv("SSG_FUSED"); return e && atoi(e) != 0; }
static inline bool fu_read_full(int fd, void *buf, size_t n)
{
	uint8_t *b = (uint8_t*)buf;
	while (n) { ssize_t r = read(fd, b, n); if (r < 0) { if (errno == EINTR) continue; return false; } if (r == 0) return false; b += r; n -= (size_t)r; }
	return true;
}
static inline bool fu_write_full(int fd, const void *buf, size_t n)
{
	const uint8_t *b = (const uint8_t*)buf;
	while (n) { ssize_t w = write(fd, b, n); if (w < 0) { if (errno == EINTR) continue; return false; } b += w; n -= (size_t)w; }
	return true;
}
static inline bool fu_write_frame(int fd, uint32_t type, const void *payload, uint64_t len)
{
	fu_frame_t f; f.type = type; f.zero = 0; f.len = len;
	return fu_write_full(fd, &f, sizeof(f)) && (!len || fu_write_full(fd, payload, (size_t)len));
}
#endif
