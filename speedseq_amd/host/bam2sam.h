/*
 * bam2sam.h -- one BAM record as the SAM line it came from: htslib's sam_format1 (/root/reference/src/samtools-1.3.1/htslib-1.3.1/sam.c:1072-1200)
 * for the field and tag types this pipeline makes (integers of every width, 'A', 'Z'; anything else is refused).  `bwa mem`'s fused hand-off
 * prints the few per cent of pairs that can reach one of samblaster's side streams this way, from the records the device made
 * (csrc/k_bam.h) -- the line is upstream mem_aln2sam's, because the record is what sam_parse1 makes of that line and the round trip is exact.
 */
#ifndef SSG_BAM2SAM_H
#define SSG_BAM2SAM_H
#include <stdint.h>
#include <string.h>
#include <string>

static inline void b2s_putl(std::string &s, long long v)
{
	char b[24]; int n = 24; unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	do { b[--n] = (char)('0' + u % 10); u /= 10; } while (u);
	if (v < 0) b[--n] = '-';
	s.append(b + n, (size_t)(24 - n));
}
/* rec: block_size-prefixed record; name(i) = contig i.  Appends the line with its '\n'; false on a record this printer does not know. */
template <class NAME> static inline bool bam_record_to_sam(const uint8_t *rec, NAME name, std::string &s)
{
	uint32_t x[9]; memcpy(x, rec, 36);
	const uint8_t *p = rec + 36, *end = rec + 4 + x[0];
	const int32_t tid = (int32_t)x[1], pos = (int32_t)x[2], mtid = (int32_t)x[6], mpos = (int32_t)x[7], isize = (int32_t)x[8];
	const uint32_t l_qname = x[3] & 0xff, mapq = x[3] >> 8 & 0xff, flag = x[4] >> 16, n_cigar = x[4] & 0xffff, l_qseq = x[5];
	if (!l_qname || p + l_qname + 4ull * n_cigar + (l_qseq + 1) / 2 + l_qseq > end) return false;
	s.append((const char*)p, l_qname - 1); s += '\t'; p += l_qname;
	b2s_putl(s, flag); s += '\t';
	if (tid >= 0) { s += name(tid); s += '\t'; } else s += "*\t";
	b2s_putl(s, (long long)pos + 1); s += '\t';
	b2s_putl(s, mapq); s += '\t';
	if (n_cigar) { for (uint32_t i = 0; i < n_cigar; ++i) { uint32_t c; memcpy(&c, p + 4 * i, 4); b2s_putl(s, c >> 4); s += "MIDNSHP=XB"[(c & 0xf) < 10 ? (c & 0xf) : 9]; } } else s += '*';
	p += 4ull * n_cigar;
	s += '\t';
	if (mtid < 0) s += "*\t"; else if (mtid == tid) s += "=\t"; else { s += name(mtid); s += '\t'; }
	b2s_putl(s, (long long)mpos + 1); s += '\t';
	b2s_putl(s, isize); s += '\t';
	if (l_qseq) {
		const size_t o = s.size(); s.resize(o + l_qseq);
		for (uint32_t i = 0; i < l_qseq; ++i) s[o + i] = "=ACMGRSVTWYHKDBN"[p[i >> 1] >> ((~i & 1) << 2) & 0xf];
		p += (l_qseq + 1) / 2;
		s += '\t';
		if (p[0] == 0xff) s += '*';
		else { const size_t q = s.size(); s.resize(q + l_qseq); for (uint32_t i = 0; i < l_qseq; ++i) s[q + i] = (char)(p[i] + 33); }
		p += l_qseq;
	} else s += "*\t*";
	while (p + 4 <= end) {
		s += '\t'; s += (char)p[0]; s += (char)p[1]; s += ':';
		const uint8_t type = p[2]; p += 3;
		if (type == 'A') { s += "A:"; s += (char)*p++; }
		else if (type == 'C') { s += "i:"; b2s_putl(s, *p); ++p; }
		else if (type == 'c') { s += "i:"; b2s_putl(s, *(const int8_t*)p); ++p; }
		else if (type == 'S') { if (p + 2 > end) return false; uint16_t v; memcpy(&v, p, 2); s += "i:"; b2s_putl(s, v); p += 2; }
		else if (type == 's') { if (p + 2 > end) return false; int16_t v; memcpy(&v, p, 2); s += "i:"; b2s_putl(s, v); p += 2; }
		else if (type == 'I') { if (p + 4 > end) return false; uint32_t v; memcpy(&v, p, 4); s += "i:"; b2s_putl(s, v); p += 4; }
		else if (type == 'i') { if (p + 4 > end) return false; int32_t v; memcpy(&v, p, 4); s += "i:"; b2s_putl(s, v); p += 4; }
		else if (type == 'Z' || type == 'H') { s += (char)type; s += ':'; while (p < end && *p) s += (char)*p++; if (p >= end) return false; ++p; }
		else return false;
	}
	s += '\n';
	return true;
}
#endif
