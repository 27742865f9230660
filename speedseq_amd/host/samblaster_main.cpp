/*
 * samblaster_main.cpp -- the `samblaster` executable speedseq.config names (reference
 * bin/speedseq.config:14, invoked at bin/speedseq:439,469):
 *   samblaster [--excludeDups] --addMateTags --maxSplitCount INT --minNonOverlap INT
 *              --splitterFile PATH --discordantFile PATH      (SAM on stdin -> SAM on stdout)
 * Host side of the drop-in boundary: SAM block parsing and the text rules (mate tags, discordant /
 * splitter extraction); duplicate marking (row a14) runs on the MI355X
 * through ssg_sbl_markdup_stream, whose signature table persists in HBM over the whole stream.
 * The two side paths are FIFOs in the reference script: they are opened before the first record is
 * read and always closed, even when empty.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include "../../include/ssgpu.h"

struct line_t {
	std::string raw; std::vector<std::string> f; std::string opt;   /* 11 mandatory fields + the rest */
	int flag, seq, pos, lclip, rclip, qalen, ralen, sqo, eqo; bool split;
};
struct opts_t { bool exclude_dups, add_mate_tags; int max_split, min_non_overlap, max_unmapped, min_indel; };

static void parse_cigar(line_t &l)
{
	l.lclip = l.rclip = l.qalen = l.ralen = 0;
	const std::string &c = l.f[5];
	if (c == "*") { l.sqo = 0; l.eqo = -1; return; }
	bool first = true; int rc = 0; size_t i = 0;
	while (i < c.size()) {
		int n = 0; while (i < c.size() && isdigit((unsigned char)c[i])) n = n * 10 + (c[i++] - '0');
		char op = c[i++];
		if (op == 'S' || op == 'H') { if (first) l.lclip += n; rc += n; }
		else {
			first = false; rc = 0;
			if (op == 'M' || op == '=' || op == 'X') { l.qalen += n; l.ralen += n; }
			else if (op == 'I') l.qalen += n;
			else if (op == 'D' || op == 'N') l.ralen += n;
		}
	}
	l.rclip = (l.qalen + l.ralen) ? rc : 0;
	l.sqo = (l.flag & 0x10) ? l.rclip : l.lclip;
	l.eqo = l.sqo + l.qalen - 1;
}

static bool parse_line(line_t &l, const std::map<std::string, int> &seqs)
{
	l.f.clear(); l.opt.clear();
	size_t p = 0;
	for (int k = 0; k < 11; ++k) {
		size_t t = l.raw.find('\t', p);
		if (t == std::string::npos) { if (k == 10) { l.f.push_back(l.raw.substr(p)); p = l.raw.size(); break; } return false; }
		l.f.push_back(l.raw.substr(p, t - p)); p = t + 1;
	}
	if (l.f.size() < 11) return false;
	if (p < l.raw.size()) l.opt = l.raw.substr(p);
	l.flag = atoi(l.f[1].c_str()); l.pos = atoi(l.f[3].c_str());
	auto it = seqs.find(l.f[2]);
	l.seq = (l.f[2] == "*" || it == seqs.end()) ? -1 : it->second;
	l.split = false;
	parse_cigar(l);
	return true;
}

static bool has_tag(const line_t &l, const char *tag)
{
	size_t p = 0;
	while (p < l.opt.size()) { if (l.opt.compare(p, 5, tag) == 0) return true; p = l.opt.find('\t', p); if (p == std::string::npos) break; ++p; }
	return false;
}

static void write_line(FILE *fp, const line_t &l, const char *suffix, const std::string &extra)
{
	fputs(l.f[0].c_str(), fp); if (suffix) fputs(suffix, fp);
	fprintf(fp, "\t%d", l.flag);
	for (int i = 2; i < 11; ++i) { fputc('\t', fp); fputs(l.f[i].c_str(), fp); }
	if (!l.opt.empty()) { fputc('\t', fp); fputs(l.opt.c_str(), fp); }
	fputs(extra.c_str(), fp);
	fputc('\n', fp);
}

static void mark_splitters(const opts_t &o, std::vector<line_t> &blk, int mask)
{
	std::vector<line_t*> arr;
	for (auto &l : blk) if (l.flag & mask) arr.push_back(&l);
	if (arr.size() < 2 || (int)arr.size() > o.max_split) return;
	for (auto *l : arr) if ((l->flag & 0x4) || l->seq < 0) return;
	std::stable_sort(arr.begin(), arr.end(), [](const line_t *a, const line_t *b) { return a->sqo < b->sqo; });
	line_t *left = arr[0];
	for (size_t i = 1; i < arr.size(); ++i) {
		line_t *right = arr[i];
		int lo = std::max(left->sqo, right->sqo), hi = std::min(left->eqo, right->eqo);
		int overlap = std::max(1 + hi - lo, 0);
		int alen1 = 1 + left->eqo - left->sqo, alen2 = 1 + right->eqo - right->sqo;
		int mno = std::min(alen1, alen2) - overlap;
		int desert = right->sqo - left->eqo - 1; bool ok = true;
		if (mno < o.min_non_overlap) ok = false;
		else if (left->seq == right->seq && (left->flag & 0x10) == (right->flag & 0x10)) {
			long long ld, rd, ins;
			if (!(left->flag & 0x10)) { ld = (long long)left->pos - left->sqo; rd = (long long)right->pos - right->sqo; ins = rd - ld; }
			else { ld = (long long)left->pos + left->ralen - 1 + left->sqo; rd = (long long)right->pos + right->ralen - 1 + right->sqo; ins = ld - rd; }
			if (desert > 0 && desert - (ins > 0 ? ins : 0) > o.max_unmapped) ok = false;
			if ((ins < 0 ? -ins : ins) < o.min_indel) ok = false;
		} else if (desert > o.max_unmapped) ok = false;
		if (ok) left->split = right->split = true;
		left = right;
	}
}

int main(int argc, char **argv)
{
	opts_t o = { false, false, 2, 20, 50, 50 };
	const char *spl_path = 0, *disc_path = 0;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--excludeDups")) o.exclude_dups = true;
		else if (!strcmp(argv[i], "--addMateTags")) o.add_mate_tags = true;
		else if (!strcmp(argv[i], "--maxSplitCount") && i + 1 < argc) o.max_split = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--minNonOverlap") && i + 1 < argc) o.min_non_overlap = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--splitterFile") && i + 1 < argc) spl_path = argv[++i];
		else if (!strcmp(argv[i], "--discordantFile") && i + 1 < argc) disc_path = argv[++i];
		else { fprintf(stderr, "[samblaster] unsupported option %s\n", argv[i]); return 1; }
	}
	FILE *spl = spl_path ? fopen(spl_path, "w") : 0, *disc = disc_path ? fopen(disc_path, "w") : 0;
	if ((spl_path && !spl) || (disc_path && !disc)) { fprintf(stderr, "[samblaster] cannot open a side file\n"); return 1; }
	ssg_sbl_state_t *st = ssg_sbl_state_new();
	std::map<std::string, int> seqs;
	const char *pg = "@PG\tID:SAMBLASTER\tVN:0.1.22-ssgpu\tCL:samblaster\n";
	const size_t CHUNK = 1u << 18;     /* pairs per GPU call */
	std::vector<std::vector<line_t> > blocks;
	char *buf = 0; size_t cap = 0; ssize_t r; bool in_header = true;
	unsigned long long n_pairs = 0, n_dups = 0, n_disc = 0, n_spl = 0;

	auto flush = [&]() {
		if (blocks.empty()) return;
		std::vector<ssg_sbl_end_t> ends(2 * blocks.size()); std::vector<uint8_t> dup(blocks.size(), 0);
		std::vector<std::pair<line_t*, line_t*> > prim(blocks.size());
		for (size_t b = 0; b < blocks.size(); ++b) {
			line_t *r1 = 0, *r2 = 0;
			for (auto &l : blocks[b]) {
				if (l.flag & (0x100 | 0x800)) continue;
				if ((l.flag & 0x40) && !r1) r1 = &l; else if ((l.flag & 0x80) && !r2) r2 = &l;
			}
			prim[b] = std::make_pair(r1, r2);
			for (int e = 0; e < 2; ++e) {
				const line_t *l = e ? r2 : r1; ssg_sbl_end_t &x = ends[2 * b + e];
				if (r1 && r2) { x.seq = ((l->flag & 0x4) || l->seq < 0) ? -1 : l->seq; x.pos = l->pos; x.flag = l->flag | (x.seq < 0 ? 0x4 : 0); x.lclip = l->lclip; x.rclip = l->rclip; x.ralen = l->ralen; }
				else { x.seq = -1; x.pos = 0; x.flag = 0x4; x.lclip = x.rclip = x.ralen = 0; }   /* unpaired block: never a duplicate */
			}
		}
		if (ssg_sbl_markdup_stream(st, (long)blocks.size(), ends.data(), dup.data())) { fprintf(stderr, "[samblaster] %s\n", ssg_last_error()); exit(1); }
		for (size_t b = 0; b < blocks.size(); ++b) {
			auto &blk = blocks[b]; line_t *r1 = prim[b].first, *r2 = prim[b].second;
			bool d = dup[b] && r1 && r2;
			if (r1 && r2) { ++n_pairs; if (d) ++n_dups; }
			std::vector<std::string> extra(blk.size());
			for (size_t i = 0; i < blk.size(); ++i) {
				line_t &l = blk[i];
				if (d) l.flag |= 0x400;
				if (o.add_mate_tags && r1 && r2) {
					const line_t *mate = (l.flag & 0x40) ? r2 : (l.flag & 0x80) ? r1 : 0;
					if (mate) {
						if (!has_tag(l, "MC:Z:")) extra[i] += "\tMC:Z:" + mate->f[5];
						if (!has_tag(l, "MQ:i:")) extra[i] += "\tMQ:i:" + mate->f[4];
					}
				}
				write_line(stdout, l, 0, extra[i]);
			}
			if (!(d && o.exclude_dups) && r1 && r2) {
				if (disc && !(r1->flag & 0x4) && !(r2->flag & 0x4) && r1->seq >= 0 && r2->seq >= 0 && !(r1->flag & 0x2)) {
					write_line(disc, *r1, 0, extra[r1 - &blk[0]]); write_line(disc, *r2, 0, extra[r2 - &blk[0]]); ++n_disc;
				}
				if (spl) {
					mark_splitters(o, blk, 0x40); mark_splitters(o, blk, 0x80);
					for (size_t i = 0; i < blk.size(); ++i) if (blk[i].split) { write_line(spl, blk[i], (blk[i].flag & 0x40) ? "_1" : "_2", extra[i]); ++n_spl; }
				}
			}
		}
		blocks.clear();
	};

	std::vector<line_t> cur;
	while ((r = getline(&buf, &cap, stdin)) > 0) {
		while (r > 0 && (buf[r-1] == '\n' || buf[r-1] == '\r')) buf[--r] = 0;
		if (in_header && buf[0] == '@') {
			if (!strncmp(buf, "@SQ", 3)) { const char *sn = strstr(buf, "\tSN:"); if (sn) { sn += 4; const char *e = strchr(sn, '\t'); std::string nm = e ? std::string(sn, e - sn) : std::string(sn); int id = (int)seqs.size(); seqs[nm] = id; } }
			fputs(buf, stdout); fputc('\n', stdout);
			if (spl) { fputs(buf, spl); fputc('\n', spl); }
			if (disc) { fputs(buf, disc); fputc('\n', disc); }
			continue;
		}
		if (in_header) { in_header = false; fputs(pg, stdout); if (spl) fputs(pg, spl); if (disc) fputs(pg, disc); }
		line_t l; l.raw.assign(buf, r);
		if (!parse_line(l, seqs)) { fprintf(stderr, "[samblaster] malformed SAM line\n"); return 1; }
		if (!cur.empty() && cur[0].f[0] != l.f[0]) { blocks.push_back(std::move(cur)); cur.clear(); if (blocks.size() >= CHUNK) flush(); }
		cur.push_back(std::move(l));
	}
	if (in_header) { fputs(pg, stdout); if (spl) fputs(pg, spl); if (disc) fputs(pg, disc); }
	if (!cur.empty()) blocks.push_back(std::move(cur));
	flush();
	if (spl) fclose(spl);
	if (disc) fclose(disc);
	ssg_sbl_state_free(st);
	free(buf);
	fprintf(stderr, "[samblaster] pairs=%llu dups=%llu discordant_pairs=%llu splitter_lines=%llu (dedup on %s)\n", n_pairs, n_dups, n_disc, n_spl, ssg_backend());
	return 0;
}
