/*
 * samblaster_main.cpp -- the `samblaster` executable speedseq.config names (reference
 * bin/speedseq.config:14, invoked at bin/speedseq:439,469):
 *   samblaster [--excludeDups] --addMateTags --maxSplitCount INT --minNonOverlap INT
 *              --splitterFile PATH --discordantFile PATH      (SAM on stdin -> SAM on stdout)
 * Host side of the drop-in boundary.  The text is scanned in place (no per-line allocation: only the six leading
 * fields are read, the rest of a line is skipped by memchr), every decision -- duplicate (a14), mate tags' source
 * line (a15), discordant (a16) and splitter (a17) membership -- is taken on the MI355X by ssg_sbl_process over the
 * numeric view of a chunk of blocks, whose duplicate set persists in HBM over the whole stream; the writer then
 * patches FLAG / appends MC,MQ / copies lines to the side streams.  The two side paths are FIFOs in the reference
 * script: they are opened before the first record is read and always closed, even when empty.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <errno.h>
#include <fcntl.h>
#include <string>
#include <vector>
#include <unordered_map>
#include <memory>
#include <thread>
#include "../../include/ssgpu.h"
#include "fastq.h"   /* chan_t */

struct out_t {           /* buffered writer on a file descriptor */
	int fd; std::vector<char> b; size_t n;
	explicit out_t(int fd_) : fd(fd_), b(4u << 20), n(0) {}
	void flush() { size_t o = 0; while (o < n) { ssize_t w = write(fd, b.data() + o, n - o); if (w < 0) { if (errno == EINTR) continue; perror("[samblaster] write"); exit(1); } o += (size_t)w; } n = 0; }
	inline void put(const char *p, size_t l) { if (n + l > b.size()) { flush(); if (l > b.size()) b.resize(l * 2); } memcpy(b.data() + n, p, l); n += l; }
	inline void putc(char c) { if (n == b.size()) flush(); b[n++] = c; }
	inline void puti(int v) { char t[16]; int k = 0; if (v == 0) t[k++] = '0'; unsigned u = (unsigned)v; char r[16]; int m = 0; while (u) { r[m++] = (char)('0' + u % 10); u /= 10; } while (m) t[k++] = r[--m]; put(t, (size_t)k); }
};

struct lrec_t {          /* one SAM line inside the chunk buffer */
	size_t off; uint32_t len;            /* line without '\n' */
	uint32_t qn_len, flag_end, cig_off, cig_len, mq_off, mq_len, opt_off;   /* relative to off; opt_off = 0: no optional fields */
};

static inline int parse_uint(const char *p, const char *e) { int v = 0; bool neg = p < e && *p == '-'; if (neg) ++p; while (p < e && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0'); return neg ? -v : v; }

static bool has_tag(const char *p, const char *e, const char *tag)
{	/* optional fields p..e: does one start with tag (5 chars)? */
	while (p < e) {
		if (e - p >= 5 && memcmp(p, tag, 5) == 0) return true;
		const char *t = (const char*)memchr(p, '\t', (size_t)(e - p));
		if (!t) break;
		p = t + 1;
	}
	return false;
}

int main(int argc, char **argv)
{
	ssg_sbl_opt_t o; ssg_sbl_opt_init(&o);
	const char *spl_path = 0, *disc_path = 0;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--excludeDups")) o.exclude_dups = 1;
		else if (!strcmp(argv[i], "--addMateTags")) o.add_mate_tags = 1;
		else if (!strcmp(argv[i], "--maxSplitCount") && i + 1 < argc) o.max_split_count = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--minNonOverlap") && i + 1 < argc) o.min_non_overlap = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--maxUnmappedBases") && i + 1 < argc) o.max_unmapped_bases = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--minIndelSize") && i + 1 < argc) o.min_indel_size = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--splitterFile") && i + 1 < argc) spl_path = argv[++i];
		else if (!strcmp(argv[i], "--discordantFile") && i + 1 < argc) disc_path = argv[++i];
		else { fprintf(stderr, "[samblaster] unsupported option %s\n", argv[i]); return 1; }
	}
	FILE *splf = spl_path ? fopen(spl_path, "w") : 0, *discf = disc_path ? fopen(disc_path, "w") : 0;
	if ((spl_path && !splf) || (disc_path && !discf)) { fprintf(stderr, "[samblaster] cannot open a side file\n"); return 1; }
	out_t out(1); out_t *spl = splf ? new out_t(fileno(splf)) : 0, *disc = discf ? new out_t(fileno(discf)) : 0;
#ifdef F_SETPIPE_SZ
	(void)fcntl(0, F_SETPIPE_SZ, 1 << 20); (void)fcntl(1, F_SETPIPE_SZ, 1 << 20);
#endif
	ssg_sbl_state_t *st = ssg_sbl_state_new();
	std::unordered_map<std::string, int> seqs;
	const char *pg = "@PG\tID:SAMBLASTER\tVN:0.1.22-ssgpu\tCL:samblaster\n";
	size_t CHUNK = 1u << 18;     /* blocks per device call */
	{ const char *e = getenv("SSG_SBL_CHUNK"); if (e && atol(e) > 0) CHUNK = (size_t)atol(e); }
	unsigned long long n_pairs = 0, n_dups = 0, n_disc = 0, n_spl = 0;

	/* a chunk of the stream: text + the in-place view of its complete blocks.  The reader thread fills and parses chunks, this
	 * thread takes the device decisions and writes; the two overlap through a bounded channel. */
	struct chunk_t {
		std::vector<char> buf; size_t have, scan; std::string header;   /* header: @-lines (+ the @PG line) that precede this chunk's records */
		std::vector<lrec_t> L; std::vector<ssg_sbl_line_t> N; std::vector<int64_t> offs;   /* offs: block starts, then the end */
		chunk_t() : buf(64u << 20), have(0), scan(0) {}
	};
	chan_t<std::unique_ptr<chunk_t> > ch(2);
	bool in_header = true; int parse_fail = 0;
	std::string last_rname; int last_seq = -1;

	auto parse_line = [&](chunk_t &C, size_t off, size_t len) -> bool {
		const char *s = C.buf.data() + off, *e = s + len, *f[12]; int nf = 0;
		f[nf++] = s;
		for (const char *p = s; nf < 7; ) { const char *t = (const char*)memchr(p, '\t', (size_t)(e - p)); if (!t) break; f[nf++] = t + 1; p = t + 1; }
		if (nf < 7) return false;
		const char *p = f[6]; int more = 0;                   /* fields 7..11 are skipped, not read */
		const char *opt = 0;
		while (more < 5) { const char *t = (const char*)memchr(p, '\t', (size_t)(e - p)); if (!t) break; p = t + 1; ++more; }
		if (more < 4) return false;                           /* fewer than 11 mandatory fields */
		if (more == 5) opt = p;
		lrec_t r; r.off = off; r.len = (uint32_t)len; r.qn_len = (uint32_t)(f[1] - 1 - s); r.flag_end = (uint32_t)(f[2] - 1 - s);
		r.mq_off = (uint32_t)(f[4] - s); r.mq_len = (uint32_t)(f[5] - 1 - f[4]); r.cig_off = (uint32_t)(f[5] - s); r.cig_len = (uint32_t)(f[6] - 1 - f[5]);
		r.opt_off = opt ? (uint32_t)(opt - s) : 0;
		ssg_sbl_line_t n; n.flag = parse_uint(f[1], f[2] - 1); n.pos = parse_uint(f[3], f[4] - 1); n.mapq = parse_uint(f[4], f[5] - 1);
		const size_t rl = (size_t)(f[3] - 1 - f[2]);
		if (rl == 1 && f[2][0] == '*') n.seq = -1;
		else if (rl == last_rname.size() && memcmp(f[2], last_rname.data(), rl) == 0) n.seq = last_seq;
		else { last_rname.assign(f[2], rl); auto it = seqs.find(last_rname); last_seq = n.seq = it == seqs.end() ? -1 : it->second; }
		n.lclip = n.rclip = n.qalen = n.ralen = 0;
		const char *c = f[5], *ce = f[6] - 1;
		if (!(ce - c == 1 && *c == '*')) {
			bool first = true; int rc = 0;
			while (c < ce) {
				int k = 0; while (c < ce && *c >= '0' && *c <= '9') k = k * 10 + (*c++ - '0');
				const char op = c < ce ? *c++ : 0;
				if (op == 'S' || op == 'H') { if (first) n.lclip += k; rc += k; }
				else { first = false; rc = 0; if (op == 'M' || op == '=' || op == 'X') { n.qalen += k; n.ralen += k; } else if (op == 'I') n.qalen += k; else if (op == 'D' || op == 'N') n.ralen += k; }
			}
			n.rclip = (n.qalen + n.ralen) ? rc : 0;
		}
		C.L.push_back(r); C.N.push_back(n);
		return true;
	};

	std::thread reader([&]() {
		std::unique_ptr<chunk_t> C(new chunk_t()); bool eof = false;
		while (!eof && !parse_fail) {
			if (C->have == C->buf.size()) C->buf.resize(C->buf.size() * 2);
			ssize_t r = read(0, C->buf.data() + C->have, C->buf.size() - C->have);
			if (r < 0) { if (errno == EINTR) continue; perror("[samblaster] read"); parse_fail = 1; break; }
			if (r == 0) { eof = true; if (C->have > C->scan && C->buf[C->have - 1] != '\n') { if (C->have == C->buf.size()) C->buf.resize(C->buf.size() + 1); C->buf[C->have++] = '\n'; } }
			else C->have += (size_t)r;
			while (C->scan < C->have) {   /* parse complete lines */
				const char *nlp = (const char*)memchr(C->buf.data() + C->scan, '\n', C->have - C->scan);
				if (!nlp) break;
				size_t len = (size_t)(nlp - (C->buf.data() + C->scan)), lo = C->scan;
				C->scan += len + 1;
				while (len && C->buf[lo + len - 1] == '\r') --len;
				if (in_header && len && C->buf[lo] == '@') {
					if (len > 3 && !memcmp(C->buf.data() + lo, "@SQ", 3)) {
						std::string ln(C->buf.data() + lo, len); size_t q = ln.find("\tSN:");
						if (q != std::string::npos) { q += 4; size_t e = ln.find('\t', q); std::string nm = ln.substr(q, e == std::string::npos ? std::string::npos : e - q); int id = (int)seqs.size(); seqs.emplace(nm, id); }
					}
					C->header.append(C->buf.data() + lo, len); C->header.push_back('\n');
					continue;
				}
				if (in_header) { in_header = false; C->header.append(pg); }
				if (!len) continue;
				if (!parse_line(*C, lo, len)) { fprintf(stderr, "[samblaster] malformed SAM line\n"); parse_fail = 1; break; }
				const size_t i = C->L.size() - 1;
				if (i == 0 || !(C->L[i - 1].qn_len == C->L[i].qn_len && memcmp(C->buf.data() + C->L[i - 1].off, C->buf.data() + C->L[i].off, C->L[i].qn_len) == 0)) C->offs.push_back((int64_t)i);
			}
			/* blocks complete so far: all but the last (it may continue in the next read) unless EOF */
			const size_t n_open = C->offs.size(), n_done = eof ? n_open : (n_open ? n_open - 1 : 0);
			if (n_done >= CHUNK || eof) {
				const int64_t cut_line = eof ? (int64_t)C->L.size() : C->offs[n_done];
				std::unique_ptr<chunk_t> nx;
				if (!eof) {   /* the open block's lines and the unparsed tail start the next chunk (re-parsed there) */
					nx.reset(new chunk_t());
					const size_t cut_byte = (size_t)cut_line < C->L.size() ? C->L[(size_t)cut_line].off : C->scan;
					const size_t keep = C->have - cut_byte;
					if (keep > nx->buf.size()) nx->buf.resize(keep * 2);
					memcpy(nx->buf.data(), C->buf.data() + cut_byte, keep); nx->have = keep;
				}
				C->L.resize((size_t)cut_line); C->N.resize((size_t)cut_line); C->offs.resize(n_done); C->offs.push_back(cut_line);
				ch.push(std::move(C));
				C = std::move(nx);
			}
		}
		ch.close();
	});

	std::vector<uint8_t> bits; std::vector<int64_t> mate;
	std::unique_ptr<chunk_t> C;
	bool wrote_header = false;
	while (ch.pop(C)) {
		chunk_t &K = *C;
		if (!K.header.empty()) { out.put(K.header.data(), K.header.size()); if (spl) spl->put(K.header.data(), K.header.size()); if (disc) disc->put(K.header.data(), K.header.size()); wrote_header = true; }
		const std::vector<int64_t> &offs = K.offs;
		const size_t n_blocks = offs.size() - 1;
		if (!n_blocks) continue;
		const std::vector<lrec_t> &L = K.L; const std::vector<ssg_sbl_line_t> &N = K.N; const std::vector<char> &buf = K.buf;
		const size_t nl = (size_t)offs[n_blocks];
		bits.resize(nl); mate.resize(nl);
		if (ssg_sbl_process(st, &o, (long)n_blocks, offs.data(), N.data(), bits.data(), mate.data())) { fprintf(stderr, "[samblaster] %s\n", ssg_last_error()); exit(1); }
		auto emit = [&](out_t &w, const lrec_t &r, int flag, bool patch_flag, const char *suffix, const lrec_t *m, bool add_mc, bool add_mq) {
			const char *s = buf.data() + r.off;
			if (!patch_flag && !suffix) w.put(s, r.len);
			else {
				w.put(s, r.qn_len); if (suffix) w.put(suffix, 2);
				w.putc('\t'); w.puti(flag);
				w.put(s + r.flag_end, r.len - r.flag_end);
			}
			if (m) {
				const char *ms = buf.data() + m->off;
				if (add_mc) { w.put("\tMC:Z:", 6); w.put(ms + m->cig_off, m->cig_len); }
				if (add_mq) { w.put("\tMQ:i:", 6); w.put(ms + m->mq_off, m->mq_len); }
			}
			w.putc('\n');
		};
		auto tags = [&](int64_t i, const lrec_t *&m, bool &add_mc, bool &add_mq) {
			const lrec_t &r = L[i];
			m = (o.add_mate_tags && mate[i] >= 0) ? &L[mate[i]] : 0; add_mc = add_mq = false;
			if (m) { const char *op = r.opt_off ? buf.data() + r.off + r.opt_off : 0, *oe = buf.data() + r.off + r.len; add_mc = !(op && has_tag(op, oe, "MC:Z:")); add_mq = !(op && has_tag(op, oe, "MQ:i:")); }
		};
		for (size_t b = 0; b < n_blocks; ++b) {
			bool paired = false, dup = false; int64_t d1 = -1, d2 = -1;
			for (int64_t i = offs[b]; i < offs[b + 1]; ++i) {
				const int bt = bits[i];
				if (mate[i] >= 0) paired = true;
				if (bt & SSG_SBL_DUP) dup = true;
				if (bt & SSG_SBL_DISC) { if (N[i].flag & 0x40) d1 = i; else d2 = i; }
				const lrec_t *m; bool add_mc, add_mq; tags(i, m, add_mc, add_mq);
				emit(out, L[i], N[i].flag | ((bt & SSG_SBL_DUP) ? 0x400 : 0), (bt & SSG_SBL_DUP) != 0, 0, m, add_mc, add_mq);
			}
			if (paired) { ++n_pairs; if (dup) ++n_dups; }
			if (disc && d1 >= 0 && d2 >= 0) {   /* upstream order: read 1's primary, then read 2's */
				for (int64_t i : { d1, d2 }) { const lrec_t *m; bool add_mc, add_mq; tags(i, m, add_mc, add_mq); emit(*disc, L[i], N[i].flag | (dup ? 0x400 : 0), dup, 0, m, add_mc, add_mq); }
				++n_disc;
			}
			if (spl) for (int64_t i = offs[b]; i < offs[b + 1]; ++i) if (bits[i] & SSG_SBL_SPLIT) {
				const lrec_t *m; bool add_mc, add_mq; tags(i, m, add_mc, add_mq);
				emit(*spl, L[i], N[i].flag | (dup ? 0x400 : 0), true, (N[i].flag & 0x40) ? "_1" : "_2", m, add_mc, add_mq); ++n_spl;
			}
		}
	}
	reader.join();
	if (parse_fail) return 1;
	(void)wrote_header;
	if (in_header) { out.put(pg, strlen(pg)); if (spl) spl->put(pg, strlen(pg)); if (disc) disc->put(pg, strlen(pg)); }
	out.flush();
	if (spl) { spl->flush(); fclose(splf); }
	if (disc) { disc->flush(); fclose(discf); }
	ssg_sbl_state_free(st);
	fprintf(stderr, "[samblaster] pairs=%llu dups=%llu discordant_pairs=%llu splitter_lines=%llu (decisions on %s)\n", n_pairs, n_dups, n_disc, n_spl, ssg_backend());
	return 0;
}
