/*
 * oracle/orc_index.c -- CPU ORACLE (test infrastructure): FM-index build / load / save / rank.
 *
 * Restates upstream lh3/bwa bntseq.c (bns_fasta2bntseq, bns_restore, bns_dump), bwtindex.c
 * (bwt_pac2bwt, bwt_bwtupdate_core), bwt.c (bwt_cal_sa, bwt_occ, bwt_2occ4, bwt_extend, bwt_sa).
 * Upstream sources are absent from /root/reference (src/bwa is an empty submodule); the on-disk
 * layout followed here is the one verified byte-exact in SURVEY.md Appendix A against
 * /root/reference/example/data/human_g1k_v37_20_42220611-42542245.fasta.{amb,ann,bwt,pac,sa},
 * and tests/test_oracle_index.py re-checks it (PINNED part of the oracle).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include <ctype.h>
#include "orc.h"

__thread uint64_t orc_cnt_extend, orc_cnt_lf, orc_cnt_sa;
uint64_t orc_tot_cells, orc_tot_extend, orc_tot_lf, orc_tot_sa;
void orc_cnt_flush(void)
{
	__atomic_fetch_add(&orc_tot_cells, orc_cnt_cells, __ATOMIC_RELAXED); __atomic_fetch_add(&orc_tot_extend, orc_cnt_extend, __ATOMIC_RELAXED);
	__atomic_fetch_add(&orc_tot_lf, orc_cnt_lf, __ATOMIC_RELAXED); __atomic_fetch_add(&orc_tot_sa, orc_cnt_sa, __ATOMIC_RELAXED);
	orc_cnt_cells = orc_cnt_extend = orc_cnt_lf = orc_cnt_sa = 0;
}

static const uint8_t nt4_table_init[5] = {'A','C','G','T','N'};
static uint8_t nt4_tab[256]; static int nt4_ready;
static void nt4_init(void)
{
	if (nt4_ready) return;
	memset(nt4_tab, 4, 256);
	for (int i = 0; i < 4; ++i) { nt4_tab[nt4_table_init[i]] = i; nt4_tab[tolower(nt4_table_init[i])] = i; }
	nt4_ready = 1;
}

/* ---------------- bns ---------------- */
int orc_bns_pos2rid(const orc_bns_t *bns, int64_t pos_f)
{	/* upstream bns_pos2rid: binary search for the contig holding forward position pos_f */
	int left, mid, right;
	if (pos_f >= bns->l_pac) return -1;
	left = 0; mid = 0; right = bns->n_seqs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= bns->anns[mid].offset) {
			if (mid == bns->n_seqs - 1) break;
			if (pos_f < bns->anns[mid+1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
int64_t orc_bns_depos(const orc_bns_t *bns, int64_t pos, int *is_rev)
{
	return (*is_rev = (pos >= bns->l_pac)) ? (bns->l_pac<<1) - 1 - pos : pos;
}
int orc_bns_intv2rid(const orc_bns_t *bns, int64_t rb, int64_t re)
{	/* upstream bns_intv2rid */
	int is_rev, rid_b, rid_e;
	if (rb < bns->l_pac && re > bns->l_pac) return -2;
	assert(rb <= re);
	rid_b = orc_bns_pos2rid(bns, orc_bns_depos(bns, rb, &is_rev));
	rid_e = rb < re ? orc_bns_pos2rid(bns, orc_bns_depos(bns, re - 1, &is_rev)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}

/* ---------------- rank ---------------- */
static inline int cnt16(uint32_t w, int c, int nsym)
{	/* number of 2-bit symbols == c among the first nsym (MSB-first) symbols of w, 0<=nsym<=16 */
	uint32_t m = ~(w ^ ((uint32_t)c * 0x55555555u));
	uint32_t t = m & (m >> 1) & 0x55555555u;
	if (nsym < 16) t &= nsym ? ~((1u << ((16 - nsym) << 1)) - 1) : 0u;
	return __builtin_popcount(t);
}
/* count of c in stored BWT [0..k] inclusive, k is the stored index (primary already removed) */
static inline uint64_t occ_raw(const orc_bwt_t *bwt, uint64_t k, int c)
{
	const uint32_t *p = bwt->bwt + ((k >> 7) << 4);
	uint64_t n = ((const uint64_t*)p)[c];
	int r = (int)(k & 127) + 1, i;
	p += 8;
	for (i = 0; i < (r >> 4); ++i) n += cnt16(p[i], c, 16);
	if (r & 15) n += cnt16(p[i], c, r & 15);
	return n;
}
uint64_t orc_bwt_occ(const orc_bwt_t *bwt, uint64_t k, int c)
{	/* upstream bwt_occ */
	if (k == bwt->seq_len) return bwt->L2[c+1] - bwt->L2[c];
	if (k == (uint64_t)(-1)) return 0;
	k -= (k >= bwt->primary);
	return occ_raw(bwt, k, c);
}
void orc_bwt_occ4(const orc_bwt_t *bwt, uint64_t k, uint64_t cnt[4])
{	/* upstream bwt_occ4 */
	if (k == (uint64_t)(-1)) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= bwt->primary);
	for (int c = 0; c < 4; ++c) cnt[c] = occ_raw(bwt, k, c);
}
static inline int bwt_B0(const orc_bwt_t *bwt, uint64_t k)
{	/* symbol at stored index k */
	const uint32_t *p = bwt->bwt + ((k >> 7) << 4) + 8;
	return p[(k & 127) >> 4] >> ((~k & 15) << 1) & 3;
}
void orc_bwt_extend(const orc_bwt_t *bwt, const orc_intv_t *ik, orc_intv_t ok[4], int is_back)
{	/* upstream bwt_extend */
	uint64_t tk[4], tl[4];
	int i;
	++orc_cnt_extend;
	orc_bwt_occ4(bwt, ik->x[!is_back] - 1, tk);
	orc_bwt_occ4(bwt, ik->x[!is_back] - 1 + ik->x[2], tl);
	for (i = 0; i != 4; ++i) {
		ok[i].x[!is_back] = bwt->L2[i] + 1 + tk[i];
		ok[i].x[2] = tl[i] - tk[i];
	}
	ok[3].x[is_back] = ik->x[is_back] + (ik->x[!is_back] <= bwt->primary && ik->x[!is_back] + ik->x[2] - 1 >= bwt->primary);
	ok[2].x[is_back] = ok[3].x[is_back] + ok[3].x[2];
	ok[1].x[is_back] = ok[2].x[is_back] + ok[2].x[2];
	ok[0].x[is_back] = ok[1].x[is_back] + ok[1].x[2];
}
static inline uint64_t bwt_invPsi(const orc_bwt_t *bwt, uint64_t k)
{	/* upstream bwt_invPsi */
	uint64_t x = k - (k > bwt->primary);
	++orc_cnt_lf;
	x = bwt_B0(bwt, x);
	x = bwt->L2[x] + orc_bwt_occ(bwt, k, (int)x);
	return k == bwt->primary ? 0 : x;
}
uint64_t orc_bwt_sa(const orc_bwt_t *bwt, uint64_t k)
{	/* upstream bwt_sa */
	uint64_t sa = 0, mask = bwt->sa_intv - 1;
	++orc_cnt_sa;
	while (k & mask) { ++sa; k = bwt_invPsi(bwt, k); }
	return sa + bwt->sa[k / bwt->sa_intv];
}

/* ---------------- construction ---------------- */
static const int32_t *g_rank; static int64_t g_k, g_n;
static int cmp_sa(const void *pa, const void *pb)
{
	int32_t a = *(const int32_t*)pa, b = *(const int32_t*)pb;
	if (g_rank[a] != g_rank[b]) return g_rank[a] < g_rank[b] ? -1 : 1;
	int32_t ra = a + g_k < g_n ? g_rank[a + g_k] : -1, rb = b + g_k < g_n ? g_rank[b + g_k] : -1;
	return ra < rb ? -1 : ra > rb;
}
static const uint64_t *g_key;
static int cmp_key(const void *pa, const void *pb)
{
	uint64_t a = g_key[*(const int32_t*)pa], b = g_key[*(const int32_t*)pb];
	return a < b ? -1 : a > b;
}
/* suffix array of T (symbols 0..3) with an implicit smallest terminator; prefix doubling */
static int32_t *build_sa(const uint8_t *T, int64_t n)
{
	int32_t *sa = malloc(n * 4), *rank = malloc(n * 4), *tmp = malloc(n * 4);
	uint64_t *key = malloc(n * 8);
	int64_t i, k;
	for (i = 0; i < n; ++i) { /* 16-symbol prefix, 3 bits per symbol, 0 = past the end */
		uint64_t x = 0;
		for (k = 0; k < 16; ++k) x = x << 3 | (i + k < n ? T[i + k] + 1 : 0);
		key[i] = x; sa[i] = (int32_t)i;
	}
	g_key = key; qsort(sa, n, 4, cmp_key);
	rank[sa[0]] = 0;
	for (i = 1; i < n; ++i) rank[sa[i]] = rank[sa[i-1]] + (key[sa[i]] != key[sa[i-1]]);
	free(key);
	for (k = 16; rank[sa[n-1]] != n - 1; k <<= 1) {
		g_rank = rank; g_k = k; g_n = n;
		/* sort only inside groups of equal rank */
		for (i = 0; i < n; ) {
			int64_t j = i + 1;
			while (j < n && rank[sa[j]] == rank[sa[i]]) ++j;
			if (j - i > 1) qsort(sa + i, j - i, 4, cmp_sa);
			i = j;
		}
		tmp[sa[0]] = 0;
		for (i = 1; i < n; ++i) tmp[sa[i]] = tmp[sa[i-1]] + (cmp_sa(&sa[i-1], &sa[i]) != 0);
		memcpy(rank, tmp, n * 4);
	}
	free(rank); free(tmp);
	return sa;
}

static orc_bwt_t *bwt_from_pac(const uint8_t *pac, int64_t l_pac)
{
	int64_t n = l_pac << 1, i;
	uint8_t *T = malloc(n);
	orc_bwt_t *bwt = calloc(1, sizeof(orc_bwt_t));
	assert(n < 0x7fffffffLL);
	for (i = 0; i < n; ++i) T[i] = orc_ref_base(pac, l_pac, i);
	int32_t *sa = build_sa(T, n);
	/* stored BWT: rows 0..n of the with-$ matrix minus the primary row */
	uint8_t *B = malloc(n);
	bwt->seq_len = n;
	uint64_t cnt[4] = {0,0,0,0};
	for (i = 0; i < n; ++i) ++cnt[T[i]];
	bwt->L2[0] = 0;
	for (i = 0; i < 4; ++i) bwt->L2[i+1] = bwt->L2[i] + cnt[i];
	/* row 0 = "$": preceding symbol is T[n-1]; row r>=1 = sa[r-1] */
	int64_t j = 0;
	B[j++] = T[n-1];
	for (i = 0; i < n; ++i) {
		if (sa[i] == 0) bwt->primary = i + 1;
		else B[j++] = T[sa[i] - 1];
	}
	assert(j == n);
	/* interleave occ checkpoints (upstream bwt_bwtupdate_core) */
	uint64_t n_occ = (n + 127) / 128 + 1;
	bwt->bwt_size = (n + 15) / 16 + n_occ * 8;
	bwt->bwt = calloc(bwt->bwt_size, 4);
	uint64_t c[4] = {0,0,0,0}, k = 0;
	for (i = 0; i < n; ++i) {
		if (i % 128 == 0) { memcpy(bwt->bwt + k, c, 32); k += 8; }
		if (i % 16 == 0) ++k;
		bwt->bwt[k-1] |= (uint32_t)B[i] << ((~i & 15) << 1);
		++c[B[i]];
	}
	memcpy(bwt->bwt + k, c, 32); k += 8;
	assert(k == bwt->bwt_size);
	/* sampled SA (upstream bwt_cal_sa, intv 32) */
	bwt->sa_intv = 32;
	bwt->n_sa = (n + 32) / 32;
	bwt->sa = malloc(bwt->n_sa * 8);
	bwt->sa[0] = (uint64_t)-1;
	for (i = 1; i < (int64_t)bwt->n_sa; ++i) bwt->sa[i] = sa[i * 32 - 1];
	free(sa); free(B); free(T);
	return bwt;
}

static const char **g_annos;   /* FASTA comments for the next orc_idx_build_mem call (upstream keeps seq->comment as anno) */
orc_idx_t *orc_idx_build_mem(int n_seqs, const char **names, const char **seqs)
{	/* upstream bns_fasta2bntseq (forward pac only, N -> lrand48()&3 with srand48(11)) + bwt build */
	orc_idx_t *idx = calloc(1, sizeof(orc_idx_t));
	orc_bns_t *bns = idx->bns = calloc(1, sizeof(orc_bns_t));
	int64_t tot = 0, m_holes = 0;
	nt4_init();
	for (int s = 0; s < n_seqs; ++s) tot += strlen(seqs[s]);
	idx->pac = calloc(tot / 4 + 2, 1);
	bns->seed = 11; srand48(bns->seed);
	bns->n_seqs = n_seqs; bns->anns = calloc(n_seqs, sizeof(orc_ann_t));
	for (int s = 0; s < n_seqs; ++s) {
		orc_ann_t *p = &bns->anns[s];
		int64_t l = strlen(seqs[s]);
		int lasts = 0;
		p->name = strdup(names[s]); p->anno = strdup(g_annos ? g_annos[s] : ""); p->gi = 0; p->len = (int32_t)l;
		p->offset = bns->l_pac; p->n_ambs = 0;
		for (int64_t i = 0; i < l; ++i) {
			int c = nt4_tab[(uint8_t)seqs[s][i]];
			if (c >= 4) { /* N: record hole, substitute random base */
				if (lasts == seqs[s][i] && bns->n_holes) ++bns->ambs[bns->n_holes-1].len;
				else {
					if (bns->n_holes == m_holes) { m_holes = m_holes ? m_holes << 1 : 16; bns->ambs = realloc(bns->ambs, m_holes * sizeof(orc_amb_t)); }
					orc_amb_t *q = &bns->ambs[bns->n_holes++];
					q->len = 1; q->offset = bns->l_pac; q->amb = seqs[s][i];
					++p->n_ambs;
				}
				lasts = seqs[s][i];
				c = lrand48() & 3;
			} else lasts = 0;
			idx->pac[bns->l_pac>>2] |= c << ((~bns->l_pac & 3) << 1);
			++bns->l_pac;
		}
	}
	idx->bwt = bwt_from_pac(idx->pac, bns->l_pac);
	return idx;
}

orc_idx_t *orc_idx_build_fasta(const char *fasta)
{
	FILE *fp = fopen(fasta, "r");
	if (!fp) return 0;
	int n = 0, m = 0; char **names = 0, **seqs = 0, **annos = 0; size_t *ls = 0, *ms = 0;
	char *line = 0; size_t cap = 0; ssize_t r;
	while ((r = getline(&line, &cap, fp)) > 0) {
		if (r > 0 && line[r-1] == '\n') line[--r] = 0;
		if (line[0] != '>') while (r > 0 && line[r-1] == '\r') line[--r] = 0;
		if (line[0] == '>') {
			if (n == m) { m = m ? m << 1 : 8; names = realloc(names, m * sizeof(char*)); annos = realloc(annos, m * sizeof(char*)); seqs = realloc(seqs, m * sizeof(char*)); ls = realloc(ls, m * sizeof(size_t)); ms = realloc(ms, m * sizeof(size_t)); }
			char *e = line + 1; while (*e && !isspace((unsigned char)*e)) ++e;
			char *c = *e ? e + 1 : e;   /* kseq_read (kseq.h:199-200): one delimiter after the name is consumed, the comment is the rest of the line ... */
			{ size_t cl = strlen(c); if (cl > 1 && c[cl-1] == '\r') c[cl-1] = 0; }   /* ... minus a trailing CR when longer than one character (ks_getuntil2, kseq.h:143) */
			annos[n] = strdup(c); *e = 0;
			names[n] = strdup(line + 1); seqs[n] = calloc(1, 1); ls[n] = 0; ms[n] = 1; ++n;
		} else if (n) {
			if (ls[n-1] + r + 1 > ms[n-1]) { ms[n-1] = (ls[n-1] + r + 1) * 2; seqs[n-1] = realloc(seqs[n-1], ms[n-1]); }
			memcpy(seqs[n-1] + ls[n-1], line, r); ls[n-1] += r; seqs[n-1][ls[n-1]] = 0;
		}
	}
	free(line); fclose(fp);
	g_annos = (const char**)annos;
	orc_idx_t *idx = orc_idx_build_mem(n, (const char**)names, (const char**)seqs);
	g_annos = 0;
	for (int i = 0; i < n; ++i) { free(names[i]); free(seqs[i]); free(annos[i]); }
	free(names); free(annos); free(seqs); free(ls); free(ms);
	return idx;
}

/* ---------------- I/O ---------------- */
static char *cat(const char *a, const char *b) { char *s = malloc(strlen(a) + strlen(b) + 1); strcpy(s, a); strcat(s, b); return s; }

int orc_idx_save(const orc_idx_t *idx, const char *prefix)
{
	const orc_bns_t *bns = idx->bns; const orc_bwt_t *bwt = idx->bwt;
	char *fn; FILE *fp;
	fn = cat(prefix, ".ann"); fp = fopen(fn, "w"); free(fn); if (!fp) return -1;
	fprintf(fp, "%lld %d %u\n", (long long)bns->l_pac, bns->n_seqs, bns->seed);
	for (int i = 0; i < bns->n_seqs; ++i) {
		const orc_ann_t *p = &bns->anns[i];
		fprintf(fp, "%d %s", p->gi, p->name);
		if (p->anno[0]) fprintf(fp, " %s\n", p->anno); else fprintf(fp, " (null)\n");
		fprintf(fp, "%lld %d %d\n", (long long)p->offset, p->len, p->n_ambs);
	}
	fclose(fp);
	fn = cat(prefix, ".amb"); fp = fopen(fn, "w"); free(fn); if (!fp) return -1;
	fprintf(fp, "%lld %d %u\n", (long long)bns->l_pac, bns->n_seqs, bns->n_holes);
	for (int i = 0; i < bns->n_holes; ++i) fprintf(fp, "%lld %d %c\n", (long long)bns->ambs[i].offset, bns->ambs[i].len, bns->ambs[i].amb);
	fclose(fp);
	fn = cat(prefix, ".pac"); fp = fopen(fn, "wb"); free(fn); if (!fp) return -1;
	fwrite(idx->pac, 1, (bns->l_pac >> 2) + ((bns->l_pac & 3) == 0 ? 0 : 1), fp);
	if (bns->l_pac % 4 == 0) { uint8_t ct = 0; fwrite(&ct, 1, 1, fp); }
	{ uint8_t ct = bns->l_pac % 4; fwrite(&ct, 1, 1, fp); }
	fclose(fp);
	fn = cat(prefix, ".bwt"); fp = fopen(fn, "wb"); free(fn); if (!fp) return -1;
	fwrite(&bwt->primary, 8, 1, fp); fwrite(bwt->L2 + 1, 8, 4, fp); fwrite(bwt->bwt, 4, bwt->bwt_size, fp);
	fclose(fp);
	fn = cat(prefix, ".sa"); fp = fopen(fn, "wb"); free(fn); if (!fp) return -1;
	uint64_t intv = bwt->sa_intv;
	fwrite(&bwt->primary, 8, 1, fp); fwrite(bwt->L2 + 1, 8, 4, fp); fwrite(&intv, 8, 1, fp); fwrite(&bwt->seq_len, 8, 1, fp);
	fwrite(bwt->sa + 1, 8, bwt->n_sa - 1, fp);
	fclose(fp);
	return 0;
}

orc_idx_t *orc_idx_load(const char *prefix)
{
	orc_idx_t *idx = calloc(1, sizeof(orc_idx_t));
	orc_bns_t *bns = idx->bns = calloc(1, sizeof(orc_bns_t));
	orc_bwt_t *bwt = idx->bwt = calloc(1, sizeof(orc_bwt_t));
	char *fn; FILE *fp; char str[8192]; long long xx; int c;
	fn = cat(prefix, ".ann"); fp = fopen(fn, "r"); free(fn); if (!fp) goto fail;
	if (fscanf(fp, "%lld%d%u", &xx, &bns->n_seqs, &bns->seed) != 3) goto fail;
	bns->l_pac = xx;
	bns->anns = calloc(bns->n_seqs, sizeof(orc_ann_t));
	for (int i = 0; i < bns->n_seqs; ++i) {
		orc_ann_t *p = &bns->anns[i]; char *q = str;
		if (fscanf(fp, "%u%8191s", &p->gi, str) != 2) goto fail;
		p->name = strdup(str);
		while ((c = fgetc(fp)) != '\n' && c != EOF) *q++ = c;
		*q = 0;
		p->anno = strdup(q - str > 1 && strcmp(str, " (null)") != 0 ? str + 1 : "");
		if (fscanf(fp, "%lld%d%d", &xx, &p->len, &p->n_ambs) != 3) goto fail;
		p->offset = xx;
	}
	fclose(fp);
	fn = cat(prefix, ".amb"); fp = fopen(fn, "r"); free(fn); if (!fp) goto fail;
	{ int n_seqs; if (fscanf(fp, "%lld%d%d", &xx, &n_seqs, &bns->n_holes) != 3) goto fail; }
	bns->ambs = calloc(bns->n_holes ? bns->n_holes : 1, sizeof(orc_amb_t));
	for (int i = 0; i < bns->n_holes; ++i) {
		if (fscanf(fp, "%lld%d%8191s", &xx, &bns->ambs[i].len, str) != 3) goto fail;
		bns->ambs[i].offset = xx; bns->ambs[i].amb = str[0];
	}
	fclose(fp);
	fn = cat(prefix, ".pac"); fp = fopen(fn, "rb"); free(fn); if (!fp) goto fail;
	idx->pac = calloc(bns->l_pac / 4 + 2, 1);
	if (fread(idx->pac, 1, bns->l_pac / 4 + 1, fp) == 0) goto fail;
	fclose(fp);
	fn = cat(prefix, ".bwt"); fp = fopen(fn, "rb"); free(fn); if (!fp) goto fail;
	fseek(fp, 0, SEEK_END); bwt->bwt_size = (ftell(fp) - 40) >> 2; fseek(fp, 0, SEEK_SET);
	bwt->bwt = calloc(bwt->bwt_size, 4);
	if (fread(&bwt->primary, 8, 1, fp) != 1 || fread(bwt->L2 + 1, 8, 4, fp) != 4) goto fail;
	if (fread(bwt->bwt, 4, bwt->bwt_size, fp) != bwt->bwt_size) goto fail;
	bwt->seq_len = bwt->L2[4];
	fclose(fp);
	fn = cat(prefix, ".sa"); fp = fopen(fn, "rb"); free(fn); if (!fp) goto fail;
	{ uint64_t hdr[7]; if (fread(hdr, 8, 7, fp) != 7) goto fail; if (hdr[0] != bwt->primary || hdr[6] != bwt->seq_len) goto fail; bwt->sa_intv = (int)hdr[5]; }
	bwt->n_sa = (bwt->seq_len + bwt->sa_intv) / bwt->sa_intv;
	bwt->sa = calloc(bwt->n_sa, 8);
	bwt->sa[0] = (uint64_t)-1;
	if (fread(bwt->sa + 1, 8, bwt->n_sa - 1, fp) != bwt->n_sa - 1) goto fail;
	fclose(fp);
	return idx;
fail:
	return 0;
}

void orc_idx_destroy(orc_idx_t *idx)
{
	if (!idx) return;
	if (idx->bns) {
		for (int i = 0; i < idx->bns->n_seqs; ++i) { free(idx->bns->anns[i].name); free(idx->bns->anns[i].anno); }
		free(idx->bns->anns); free(idx->bns->ambs); free(idx->bns);
	}
	if (idx->bwt) { free(idx->bwt->bwt); free(idx->bwt->sa); free(idx->bwt); }
	free(idx->pac); free(idx);
}
