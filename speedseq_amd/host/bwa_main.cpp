/*
 * bwa_main.cpp -- the `bwa` executable speedseq.config names (reference bin/speedseq.config:13,
 * invoked at bin/speedseq:389,438,468): `bwa index <ref>` and
 * `bwa mem -t INT [-p] [-I F[,F[,I[,I]]]] -R STR [-C] <ref> <fq1> [fq2]` -> SAM on stdout.
 * Host side of the drop-in boundary: FASTQ parsing, upstream's batch boundaries and SAM printing;
 * all alignment work happens on the MI355X through libssgpu's C ABI (include/ssgpu.h).
 * Mirrors upstream main_mem() (fastmap.c): bseq_read(chunk_size * n_threads) -> mem_process_seqs
 * -> fputs(sam).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <string>
#include <vector>
#include <zlib.h>
#include <unistd.h>
#include "../../include/ssgpu.h"

static void die(const char *what) { fprintf(stderr, "[bwa] %s: %s\n", what, ssg_last_error()); exit(1); }

struct fq_t {
	gzFile fp; std::vector<char> buf;
	bool getline(std::string &out)
	{
		out.clear();
		if (buf.empty()) buf.resize(1 << 16);
		for (;;) {
			if (!gzgets(fp, buf.data(), (int)buf.size())) return !out.empty();
			out.append(buf.data());
			if (!out.empty() && out.back() == '\n') break;
		}
		while (!out.empty() && (out.back() == '\n' || out.back() == '\r')) out.pop_back();
		return true;
	}
};
struct read_t { std::string name, comment, qual; std::vector<uint8_t> seq; bool has_qual; };

static uint8_t nt4(int c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }

/* one FASTQ/FASTA record with kseq semantics (name up to first blank, optional comment);
 * upstream bseq trims a trailing /1 or /2 */
static int read1(fq_t &f, read_t &r, bool keep_comment)
{
	std::string l;
	do { if (!f.getline(l)) return -1; } while (l.empty());
	if (l[0] != '@' && l[0] != '>') return -2;
	bool is_fq = l[0] == '@';
	size_t e = 1; while (e < l.size() && !isspace((unsigned char)l[e])) ++e;
	r.name = l.substr(1, e - 1);
	while (e < l.size() && isspace((unsigned char)l[e])) ++e;
	r.comment = keep_comment ? l.substr(e) : std::string();
	if (r.name.size() > 2 && r.name[r.name.size() - 2] == '/' && isdigit((unsigned char)r.name.back())) r.name.resize(r.name.size() - 2);
	if (!f.getline(l)) return -2;
	r.seq.resize(l.size());
	for (size_t i = 0; i < l.size(); ++i) r.seq[i] = nt4(l[i]);
	r.has_qual = false;
	if (is_fq) {
		if (!f.getline(l) || l.empty() || l[0] != '+') return -2;
		if (!f.getline(l) || l.size() != r.seq.size()) return -2;
		r.qual = l; r.has_qual = true;
	}
	return 0;
}

static std::string unescape(const char *s)
{	/* upstream bwa_set_rg: "\t" -> TAB */
	std::string o;
	for (const char *p = s; *p; ++p) {
		if (*p == '\\' && p[1] == 't') { o += '\t'; ++p; }
		else if (*p == '\\' && p[1] == 'n') { o += '\n'; ++p; }
		else if (*p == '\\' && p[1] == '\\') { o += '\\'; ++p; }
		else o += *p;
	}
	return o;
}

static int main_index(int argc, char **argv)
{	/* upstream bwa_index (bwtindex.c): `bwa index [-p prefix] [-a algo] <in.fasta>`; the FM-index is built on the GPU by libssgpu (ssg_index_build_fasta) */
	const char *prefix = 0; int ai = 1;
	for (; ai < argc && argv[ai][0] == '-' && argv[ai][1]; ++ai) {
		if (!strcmp(argv[ai], "-p") && ai + 1 < argc) prefix = argv[++ai];
		else if (!strcmp(argv[ai], "-a") && ai + 1 < argc) ++ai;          /* construction algorithm: irrelevant here, same bytes */
		else if (!strcmp(argv[ai], "-b") && ai + 1 < argc) ++ai;
		else if (!strcmp(argv[ai], "-6")) ;
		else { fprintf(stderr, "[bwa] index: unsupported option %s\n", argv[ai]); return 1; }
	}
	if (ai >= argc) { fprintf(stderr, "usage: bwa index [-p prefix] <ref.fa>\n"); return 1; }
	ssg_index_t *idx;
	if (ssg_index_build_fasta(argv[ai], &idx)) die("index construction failed");
	if (ssg_index_save(idx, prefix ? prefix : argv[ai])) die("writing the index failed");
	fprintf(stderr, "[bwa] index: %lld bp in %d sequence(s) indexed on %s\n", (long long)ssg_index_l_pac(idx), ssg_index_n_ctg(idx), ssg_backend());
	ssg_index_destroy(idx);
	return 0;
}

static int main_mem(int argc, char **argv)
{
	ssg_mem_opt_t opt; ssg_mem_opt_init(&opt);
	bool interleaved = false, keep_comment = false; std::string rg; char rg_id[256] = "";
	ssg_pestat_t pes0[4], *pes = 0;
	int ai = 1;
	for (; ai < argc && argv[ai][0] == '-' && argv[ai][1]; ++ai) {
		const char *a = argv[ai];
		if (!strcmp(a, "-p")) interleaved = true;
		else if (!strcmp(a, "-C")) keep_comment = true;
		else if (!strcmp(a, "-M")) ;
		else if (!strcmp(a, "-t") && ai + 1 < argc) opt.n_threads = atoi(argv[++ai]);
		else if (!strcmp(a, "-R") && ai + 1 < argc) rg = unescape(argv[++ai]);
		else if (!strcmp(a, "-I") && ai + 1 < argc) { /* upstream main_mem -I: FR orientation only */
			char *p; pes = pes0; memset(pes0, 0, sizeof(pes0));
			pes0[0].failed = pes0[2].failed = pes0[3].failed = 1;
			pes0[1].avg = strtod(argv[++ai], &p);
			pes0[1].std = pes0[1].avg * .1;
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0[1].std = strtod(p + 1, &p);
			pes0[1].high = (int)(pes0[1].avg + 4. * pes0[1].std + .499);
			pes0[1].low  = (int)(pes0[1].avg - 4. * pes0[1].std + .499);
			if (pes0[1].low < 1) pes0[1].low = 1;
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0[1].high = (int)(strtod(p + 1, &p) + .499);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0[1].low  = (int)(strtod(p + 1, &p) + .499);
		} else { fprintf(stderr, "[bwa] unsupported option %s\n", a); return 1; }
	}
	if (argc - ai < 2) { fprintf(stderr, "usage: bwa mem [-t INT] [-p] [-I ...] [-R STR] [-C] <ref> <fq1> [fq2]\n"); return 1; }
	if (opt.n_threads < 1) opt.n_threads = 1;
	if (!rg.empty()) {
		if (rg.compare(0, 3, "@RG") != 0) { fprintf(stderr, "[E::bwa_set_rg] the read group line is not started with @RG\n"); return 1; }
		size_t p = rg.find("\tID:");
		if (p == std::string::npos) { fprintf(stderr, "[E::bwa_set_rg] no ID at the read group line\n"); return 1; }
		size_t i = 0; for (p += 4; p < rg.size() && rg[p] != '\t' && rg[p] != '\n' && i < 255; ++p) rg_id[i++] = rg[p];
		rg_id[i] = 0;
	}
	ssg_index_t *idx;
	if (ssg_index_load(argv[ai], &idx)) die("fail to load the index");
	fq_t f1, f2; f1.fp = gzopen(argv[ai + 1], "r"); f2.fp = 0;
	if (!f1.fp) { fprintf(stderr, "[bwa] fail to open %s\n", argv[ai + 1]); return 1; }
	if (argc - ai >= 3) { f2.fp = gzopen(argv[ai + 2], "r"); if (!f2.fp) { fprintf(stderr, "[bwa] fail to open %s\n", argv[ai + 2]); return 1; } }
	if (!interleaved && !f2.fp) { fprintf(stderr, "[bwa] single-end input is not supported: speedseq align is paired-end\n"); return 1; }
	/* header: upstream bwa_print_sam_hdr + @PG */
	for (int i = 0; i < ssg_index_n_ctg(idx); ++i) printf("@SQ\tSN:%s\tLN:%d\n", ssg_index_name(idx, i), ssg_index_len(idx, i));
	if (!rg.empty()) printf("%s\n", rg.c_str());
	{ std::string cl = "bwa"; for (int i = 0; i < argc; ++i) { cl += ' '; cl += argv[i]; } printf("@PG\tID:bwa\tPN:bwa\tVN:0.7.12-ssgpu\tCL:%s\n", cl.c_str()); }
	const int64_t chunk = (int64_t)opt.chunk_size * opt.n_threads;
	const size_t max_pairs_per_call = 1u << 20;     /* several upstream batches go to the GPU at once */
	int64_t id0 = 0; bool eof = false;
	while (!eof) {
		std::vector<read_t> reads; std::vector<int32_t> pair_batch; int n_batches = 0;
		while (!eof && reads.size() / 2 < max_pairs_per_call) { /* upstream bseq_read: one batch */
			int64_t size = 0; size_t n0 = reads.size();
			for (;;) {
				read_t a, b; int rc = read1(f1, a, keep_comment);
				if (rc == -1) { eof = true; break; }
				if (rc < 0 || read1(f2.fp ? f2 : f1, b, keep_comment) < 0) { fprintf(stderr, "[bwa] truncated or malformed FASTQ (paired reads expected)\n"); return 1; }
				if (a.name != b.name) { fprintf(stderr, "[mem_sam_pe] paired reads have different names: \"%s\", \"%s\"\n", a.name.c_str(), b.name.c_str()); return 1; }
				size += (int64_t)a.seq.size() + (int64_t)b.seq.size();
				reads.push_back(std::move(a)); reads.push_back(std::move(b));
				if (size >= chunk) break;
			}
			if (reads.size() > n0) { for (size_t p = n0 / 2; p < reads.size() / 2; ++p) pair_batch.push_back(n_batches); ++n_batches; }
		}
		if (reads.empty()) break;
		const int n = (int)reads.size();
		std::vector<int64_t> off(n + 1); std::vector<uint8_t> seq; std::vector<const char*> names(n), quals(n), comments(n);
		off[0] = 0;
		for (int i = 0; i < n; ++i) {
			seq.insert(seq.end(), reads[i].seq.begin(), reads[i].seq.end()); off[i + 1] = (int64_t)seq.size();
			names[i] = reads[i].name.c_str(); quals[i] = reads[i].has_qual ? reads[i].qual.c_str() : 0; comments[i] = reads[i].comment.empty() ? 0 : reads[i].comment.c_str();
		}
		ssg_pe_result_t *res; char *sam; std::vector<int64_t> sam_off(n + 1);
		if (ssg_mem_process_pairs(idx, &opt, n / 2, seq.data(), off.data(), pair_batch.data(), n_batches, id0, pes, &res)) die("alignment failed");
		if (ssg_sam_format(idx, &opt, res, n / 2, names.data(), seq.data(), off.data(), quals.data(), comments.data(), rg_id, &sam, sam_off.data())) die("SAM formatting failed");
		fwrite(sam, 1, (size_t)sam_off[n], stdout);
		const ssg_pestat_t *pp = ssg_pe_pes(res);
		fprintf(stderr, "[bwa] processed %d reads in %d upstream batch(es) on %s; FR insert (first batch): failed=%d low=%d high=%d avg=%.2f std=%.2f\n",
		        n, n_batches, ssg_backend(), pp[1].failed, pp[1].low, pp[1].high, pp[1].avg, pp[1].std);
		ssg_free(sam); ssg_pe_result_free(res);
		id0 += n / 2;
	}
	gzclose(f1.fp); if (f2.fp) gzclose(f2.fp);
	ssg_index_destroy(idx);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: bwa <index|mem> ...  (libssgpu %s, %s)\n", ssg_version(), ssg_backend()); return 1; }
	if (!strcmp(argv[1], "index")) return main_index(argc - 1, argv + 1);
	if (!strcmp(argv[1], "mem")) return main_mem(argc - 1, argv + 1);
	fprintf(stderr, "[bwa] unknown command %s\n", argv[1]);
	return 1;
}
