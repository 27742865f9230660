/*
 * bamkit -- the four helpers `speedseq realign` drives (reference bin/speedseq:1886-1968), as one native executable that answers
 * to their names: bamtofastq.py, bamheadrg.py, bamcleanheader.py, bamlibs.py.  Host code only (no device
 * work): BAM decoding on the threaded BGZF reader of bamio.h, text out.
 *
 * Upstream these are the python scripts of hall-lab/bamkit (an empty submodule in the reference tree: src/bamkit), so this is a
 * restatement of their documented behaviour, anchored on how the reference script uses them:
 *   bamcleanheader.py in.bam [...]            > header.txt     one merged header ("sloppy": @HD and @SQ of the first file, the
 *                                                              distinct @RG lines of all files; @PG / @CO dropped)
 *   bamlibs.py -S header.txt                                   one line per library (LB of the @RG lines, in order of first
 *                                                              appearance): its read-group ids joined by commas
 *   bamtofastq.py [-r id[,id...]] [-n] in.bam [...]            interleaved FASTQ of the pairs whose two primary records are in the
 *                                                              input (any order: a coordinate-sorted BAM is the normal case),
 *                                                              each read in its original orientation, `RG:Z:<id>` as the comment
 *                                                              that `bwa mem -C` copies into its SAM record; -n: names become
 *                                                              a running number
 *   bamheadrg.py -d header.txt [-r id[,id...]] < in.sam        the SAM stream with the donor's @RG lines (all, or the listed ids)
 *                                                              added to its header
 * The reference script runs them as `$PYTHON <script> args`; speedseq.config names bin/pyrun as $PYTHON (it executes its first
 * argument and answers the script's module checks).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <algorithm>
#include <functional>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>
#include "bamio.h"

static void die(const std::string &m) { fprintf(stderr, "[bamkit] %s\n", m.c_str()); exit(1); }
static int open_in(const char *p) { if (!strcmp(p, "/dev/stdin") || !strcmp(p, "-")) return 0; int fd = open(p, O_RDONLY); if (fd < 0) die(std::string("cannot open ") + p); return fd; }
static std::vector<std::string> split(const std::string &s, char c)
{
	std::vector<std::string> v; size_t a = 0;
	for (;;) { size_t b = s.find(c, a); if (b == std::string::npos) { v.push_back(s.substr(a)); break; } v.push_back(s.substr(a, b - a)); a = b + 1; }
	return v;
}
static std::vector<std::string> header_lines(const std::string &text)
{
	std::vector<std::string> v;
	for (const std::string &l : split(text, '\n')) if (!l.empty()) v.push_back(l);
	return v;
}
static std::string tag_of(const std::string &line, const char *tag)
{	/* value of TAG: in a tab-separated header line, "" if absent */
	const std::vector<std::string> f = split(line, '\t');
	for (size_t i = 1; i < f.size(); ++i) if (f[i].size() >= 3 && f[i][0] == tag[0] && f[i][1] == tag[1] && f[i][2] == ':') return f[i].substr(3);
	return "";
}
static std::string read_file(const char *path)
{
	FILE *fp = fopen(path, "rb"); if (!fp) die(std::string("cannot open ") + path);
	std::string s; char buf[65536]; size_t n;
	while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) s.append(buf, n);
	fclose(fp);
	return s;
}

/* ---------------- bamcleanheader ---------------- */
static int main_cleanheader(int argc, char **argv)
{
	if (argc < 1) { fprintf(stderr, "usage: bamcleanheader.py <in.bam> [in2.bam ...]\n"); return 1; }
	std::string out; std::unordered_set<std::string> seen_rg;
	for (int i = 0; i < argc; ++i) {
		bgzf_in_t in(open_in(argv[i]), 2); bam_hdr_t h;
		if (!hdr_read(in, h)) die(std::string(argv[i]) + ": not a BAM file");
		for (const std::string &l : header_lines(h.text)) {
			const std::string k = l.substr(0, 3);
			if (k == "@HD" || k == "@SQ") { if (i == 0) out += l + "\n"; }
			else if (k == "@RG") { if (seen_rg.insert(l).second) out += l + "\n"; }
		}
		if (i == 0 && h.text.find("@SQ") == std::string::npos)   /* a header without text: the binary reference list speaks */
			for (size_t t = 0; t < h.names.size(); ++t) out += "@SQ\tSN:" + h.names[t] + "\tLN:" + std::to_string(h.lens[t]) + "\n";
		close(in.fd);
	}
	fputs(out.c_str(), stdout);
	return 0;
}

/* ---------------- bamlibs ---------------- */
static int main_libs(int argc, char **argv)
{
	const char *path = 0; bool is_sam = false;
	for (int i = 0; i < argc; ++i) { if (!strcmp(argv[i], "-S")) is_sam = true; else path = argv[i]; }
	if (!path) { fprintf(stderr, "usage: bamlibs.py [-S] <header.sam | in.bam>\n"); return 1; }
	std::string text;
	if (is_sam) text = read_file(path);
	else { bgzf_in_t in(open_in(path), 2); bam_hdr_t h; if (!hdr_read(in, h)) die(std::string(path) + ": not a BAM file"); text = h.text; }
	std::vector<std::string> libs; std::unordered_map<std::string, std::vector<std::string> > ids;
	for (const std::string &l : header_lines(text)) {
		if (l.compare(0, 3, "@RG")) continue;
		const std::string id = tag_of(l, "ID"), lb = tag_of(l, "LB");
		if (id.empty()) continue;
		if (!ids.count(lb)) libs.push_back(lb);
		std::vector<std::string> &v = ids[lb];
		if (std::find(v.begin(), v.end(), id) == v.end()) v.push_back(id);
	}
	for (const std::string &lb : libs) {
		const std::vector<std::string> &v = ids[lb];
		for (size_t i = 0; i < v.size(); ++i) printf("%s%s", i ? "," : "", v[i].c_str());
		printf("\n");
	}
	return 0;
}

/* ---------------- bamheadrg ---------------- */
static int main_headrg(int argc, char **argv)
{
	const char *donor = 0; std::set<std::string> want;
	for (int i = 0; i < argc; ++i) {
		if (!strcmp(argv[i], "-d") && i + 1 < argc) donor = argv[++i];
		else if (!strcmp(argv[i], "-r") && i + 1 < argc) { for (const std::string &r : split(argv[++i], ',')) if (!r.empty()) want.insert(r); }
		else if (!strcmp(argv[i], "-h")) { fprintf(stderr, "usage: bamheadrg.py -d <donor header.sam> [-r id[,id...]] < in.sam > out.sam\n"); return 1; }
	}
	if (!donor) die("bamheadrg: -d <donor header> is required");
	std::string inject;
	for (const std::string &l : header_lines(read_file(donor)))
		if (!l.compare(0, 3, "@RG") && (want.empty() || want.count(tag_of(l, "ID")))) inject += l + "\n";
	/* the stream: header lines pass; the donor's @RG lines go in before the first record (after bwa's @SQ / @PG), minus ids the
	 * stream's own header already has; then bytes are copied through */
	std::vector<char> buf((size_t)1 << 22); std::string carry; bool in_header = true; std::unordered_set<std::string> own;
	auto flush_inject = [&]() {
		for (const std::string &l : header_lines(inject)) if (!own.count(tag_of(l, "ID"))) { fputs(l.c_str(), stdout); fputc('\n', stdout); }
	};
	ssize_t n;
	while ((n = read(0, buf.data(), buf.size())) != 0) {
		if (n < 0) { if (errno == EINTR) continue; die("read error on stdin"); }
		if (!in_header) { fwrite(buf.data(), 1, (size_t)n, stdout); continue; }
		carry.append(buf.data(), (size_t)n);
		size_t a = 0;
		while (in_header) {
			if (a >= carry.size()) break;
			if (carry[a] != '@') { flush_inject(); in_header = false; break; }
			const size_t b = carry.find('\n', a);
			if (b == std::string::npos) break;                    /* an incomplete header line: wait for more */
			const std::string l = carry.substr(a, b - a);
			if (!l.compare(0, 3, "@RG")) own.insert(tag_of(l, "ID"));
			fwrite(carry.data() + a, 1, b + 1 - a, stdout);
			a = b + 1;
		}
		if (!in_header) { fwrite(carry.data() + a, 1, carry.size() - a, stdout); carry.clear(); }
		else carry.erase(0, a);
	}
	if (in_header) { fputs(carry.c_str(), stdout); flush_inject(); }   /* a header-only stream */
	fflush(stdout);
	return 0;
}

/* ---------------- bamtofastq ---------------- */
struct half_t { std::string seq, qual, rg; bool first; };   /* a read waiting for its mate: already in its original orientation */

static void decode_read(const uint8_t *rec, int32_t len, half_t &h, std::string &name, uint32_t &flag, bool &hard_clipped)
{
	bam_core_t c; memcpy(&c, rec, 32);
	const uint32_t l_qname = c.bin_mq_nl & 0xff, n_cigar = c.flag_nc & 0xffff;
	flag = c.flag_nc >> 16;
	name.assign((const char*)rec + 32, l_qname ? l_qname - 1 : 0);
	const uint8_t *p = rec + 32 + l_qname;
	hard_clipped = false;
	for (uint32_t i = 0; i < n_cigar; ++i) { uint32_t v; memcpy(&v, p + 4 * i, 4); if ((v & 0xf) == 5) hard_clipped = true; }
	p += 4 * n_cigar;
	const int32_t l = c.l_qseq;
	h.seq.resize((size_t)l); h.qual.resize((size_t)l);
	for (int32_t i = 0; i < l; ++i) h.seq[(size_t)i] = SEQ_NT16[(p[i >> 1] >> ((~i & 1) << 2)) & 15];
	p += (l + 1) >> 1;
	const bool no_qual = l > 0 && p[0] == 0xff;
	for (int32_t i = 0; i < l; ++i) h.qual[(size_t)i] = no_qual ? 'I' : (char)(p[i] + 33);
	p += l;
	if (flag & 0x10) {   /* stored reverse-complemented: back to the read as sequenced */
		std::reverse(h.seq.begin(), h.seq.end()); std::reverse(h.qual.begin(), h.qual.end());
		for (char &ch : h.seq) { switch (ch) { case 'A': ch = 'T'; break; case 'C': ch = 'G'; break; case 'G': ch = 'C'; break; case 'T': ch = 'A'; break;
			case 'M': ch = 'K'; break; case 'K': ch = 'M'; break; case 'R': ch = 'Y'; break; case 'Y': ch = 'R'; break; case 'V': ch = 'B'; break; case 'B': ch = 'V'; break;
			case 'H': ch = 'D'; break; case 'D': ch = 'H'; break; default: break; } }
	}
	h.first = (flag & 0x40) != 0;
	h.rg.clear();
	const uint8_t *e = rec + len;   /* aux: find RG:Z */
	while (p + 3 <= e) {
		const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2]; p += 3;
		size_t sz = 0;
		switch (ty) {
			case 'A': case 'c': case 'C': sz = 1; break;
			case 's': case 'S': sz = 2; break;
			case 'i': case 'I': case 'f': sz = 4; break;
			case 'Z': case 'H': { const uint8_t *q = (const uint8_t*)memchr(p, 0, (size_t)(e - p)); if (!q) return; if (t0 == 'R' && t1 == 'G' && ty == 'Z') h.rg.assign((const char*)p, (size_t)(q - p)); sz = (size_t)(q - p) + 1; } break;
			case 'B': { if (p + 5 > e) return; const char st = (char)p[0]; uint32_t cnt; memcpy(&cnt, p + 1, 4); const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; sz = 5 + es * cnt; } break;
			default: return;
		}
		p += sz;
	}
}

static int main_tofastq(int argc, char **argv)
{
	std::set<std::string> want; bool rename = false; std::vector<const char*> files;
	for (int i = 0; i < argc; ++i) {
		if (!strcmp(argv[i], "-r") && i + 1 < argc) { for (const std::string &r : split(argv[++i], ',')) if (!r.empty()) want.insert(r); }
		else if (!strcmp(argv[i], "-n")) rename = true;
		else if (!strcmp(argv[i], "-h")) { fprintf(stderr, "usage: bamtofastq.py [-r id[,id...]] [-n] <in.bam> [in2.bam ...] > interleaved.fq\n"); return 1; }
		else files.push_back(argv[i]);
	}
	if (files.empty()) { fprintf(stderr, "usage: bamtofastq.py [-r id[,id...]] [-n] <in.bam> [in2.bam ...] > interleaved.fq\n"); return 1; }
	const int threads = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
	std::unordered_map<std::string, half_t> waiting;
	std::string out; out.reserve((size_t)1 << 22);
	uint64_t counter = 0;
	auto emit = [&](const std::string &name, const half_t &h, int which) {
		out += '@'; out += name; out += '/'; out += (char)('0' + which);
		if (!h.rg.empty()) { out += " RG:Z:"; out += h.rg; }
		out += '\n'; out += h.seq; out += "\n+\n"; out += h.qual; out += '\n';
	};
	std::vector<uint8_t> rec; std::string name; half_t cur;
	for (const char *path : files) {
		bgzf_in_t in(open_in(path), threads); bam_hdr_t hd;
		if (!hdr_read(in, hd)) die(std::string(path) + ": not a BAM file");
		for (;;) {
			int32_t len;
			const size_t g = in.get(&len, 4);
			if (g == 0) break;
			if (g != 4 || len < 32) die(std::string(path) + ": truncated BAM record");
			rec.resize((size_t)len);
			if (in.get(rec.data(), (size_t)len) != (size_t)len) die(std::string(path) + ": truncated BAM record");
			uint32_t flag; bool hard;
			decode_read(rec.data(), len, cur, name, flag, hard);
			if ((flag & 0x900) || hard || !(flag & 1)) continue;       /* primary records of paired reads only, whole reads */
			if (!want.empty() && !want.count(cur.rg)) continue;
			auto it = waiting.find(name);
			if (it == waiting.end()) { waiting.emplace(name, cur); continue; }
			if (it->second.first == cur.first) continue;                /* the same end twice: keep waiting for the other */
			const half_t &a = cur.first ? cur : it->second, &b = cur.first ? it->second : cur;
			const std::string nm = rename ? std::to_string(counter) : name;
			emit(nm, a, 1); emit(nm, b, 2);
			++counter;
			waiting.erase(it);
			if (out.size() > ((size_t)1 << 22) - 65536) { io_write_all(1, out.data(), out.size()); out.clear(); }
		}
		close(in.fd);
	}
	if (!out.empty()) io_write_all(1, out.data(), out.size());
	if (!waiting.empty()) fprintf(stderr, "[bamtofastq] %zu reads without their mate in the input were left out\n", waiting.size());
	return 0;
}

int main(int argc, char **argv)
{
	std::string me = argv[0]; const size_t sl = me.rfind('/'); if (sl != std::string::npos) me = me.substr(sl + 1);
	int a0 = 1;
	if (me.compare(0, 6, "bamkit") == 0) {   /* bamkit <tool> args */
		if (argc < 2) { fprintf(stderr, "usage: bamkit <bamtofastq|bamheadrg|bamcleanheader|bamlibs> [args]\n"); return 1; }
		me = argv[1]; a0 = 2;
	}
	if (me.compare(0, 10, "bamtofastq") == 0) return main_tofastq(argc - a0, argv + a0);
	if (me.compare(0, 9, "bamheadrg") == 0) return main_headrg(argc - a0, argv + a0);
	if (me.compare(0, 14, "bamcleanheader") == 0) return main_cleanheader(argc - a0, argv + a0);
	if (me.compare(0, 7, "bamlibs") == 0) return main_libs(argc - a0, argv + a0);
	fprintf(stderr, "[bamkit] unknown tool '%s'\n", me.c_str());
	return 1;
}
