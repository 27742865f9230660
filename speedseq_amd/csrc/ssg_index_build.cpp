/*
 * ssg_index_build.cpp -- `bwa index` on the MI355X: host orchestration of k_index.h and the index
 * file writer.  Replaces upstream bwa_idx_build (bwtindex.c) as invoked by the reference at
 * /root/reference/bin/speedseq:386-391; output bytes follow SURVEY.md Appendix A (verified against
 * the reference's bundled example index, tests/test_index_build.py).
 * Compiled by hipcc for gfx950 (product) or by g++ with -DSSG_EMU against tests/emu (CPU tests).
 */
#include <vector>
#include <string>
#include <algorithm>
#include <zlib.h>
#include "ssg_rt.h"
#include "k_index.h"
#include "ssg_prim.h"
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"

#define CHK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

/* build-time arrays: straight from / back to the driver, never through the arena */
template <class T> struct rbuf {
	T *p; size_t n;
	rbuf() : p(0), n(0) {}
	~rbuf() { rt_free_raw(p); }
	rbuf(const rbuf&) = delete; rbuf &operator=(const rbuf&) = delete;
	bool alloc(size_t n_) { rt_free_raw(p); n = n_; p = (T*)rt_malloc_raw((n_ ? n_ : 1) * sizeof(T)); return p != 0; }
	void release() { rt_free_raw(p); p = 0; n = 0; }
	T *take() { T *q = p; p = 0; n = 0; return q; }
};
#define RALLOC(b, cnt) do { if (!(b).alloc(cnt)) { ssg_err_msg = "index construction: device allocation failed (" #b ")"; return SSG_ENOMEM; } } while (0)

static int idx_verbose() { const char *e = getenv("SSG_INDEX_VERBOSE"); return e && *e ? atoi(e) : 0; }
/* grid of 256-lane workgroups for n items; capped (the kernels are grid-stride): a HIP launch must stay below 2^32 threads */
static inline long nblk256(int64_t n)
{
	static long cap = 0;
	if (!cap) { const char *e = getenv("SSG_INDEX_MAX_WG"); cap = e && atol(e) > 0 ? atol(e) : (1L << 20); }   /* the tests shrink it to exercise the stride */
	const int64_t b = (n + 255) / 256;
	return (long)(b < cap ? b : cap);
}

/* suffix array of T[0..n) (implicit smallest terminator) into d_SA; d_T is zero-padded by >= 64 bytes */
static int build_suffix_array(const uint8_t *d_T, int64_t n, uint64_t *d_SA)
{
	const char *pe = getenv("SSG_INDEX_BUCKET_P");
	int p = 0;
	if (pe && *pe) p = atoi(pe);
	else while (p < 8 && (n >> (2 * p)) > (1LL << 28)) ++p;
	if (p < 0 || p > 8) { ssg_err_msg = "SSG_INDEX_BUCKET_P out of range"; return SSG_EINVAL; }
	const uint32_t n_bucket = 1u << (2 * p);
	rbuf<uint64_t> rank;
	RALLOC(rank, (size_t)n);
	const int64_t n_wg = (n + SSG_IDX_BK_PER_WG - 1) / SSG_IDX_BK_PER_WG;
	rbuf<uint32_t> wg_cnt; rbuf<uint64_t> wg_off;
	RALLOC(wg_cnt, (size_t)n_wg + 1); RALLOC(wg_off, (size_t)n_wg + 2);
	struct chunk_t { uint64_t *pos, *grp; uint64_t m; };
	std::vector<chunk_t> chunks;
	auto free_chunks = [&]() { for (auto &c : chunks) { rt_free_raw(c.pos); rt_free_raw(c.grp); } chunks.clear(); };
	uint64_t base = 0, n_pend = 0;
	int rc = 0;
	/* ---- round 1: per bucket of equal p-symbol prefix, sort by the next 32 symbols ---- */
	for (uint32_t b = 0; b < n_bucket && !rc; ++b) {
		uint64_t m = 0;
		/* the bucket's suffixes in text order: per-workgroup counts straight from the text, their prefix sums, the scatter (k_index.h) */
		SSG_LAUNCH(ssg_k_idx_bucket_count, n_wg, 256, 0, d_T, n, p, b, wg_cnt.p);
		if ((rc = prim_exsum_u32_u64(wg_cnt.p, wg_off.p, n_wg))) break;
		if ((rc = rt_d2h(&m, wg_off.p + n_wg, 8))) break;
		if (m == 0) continue;
		rbuf<uint64_t> pos, key, ps, ks, grp; rbuf<int64_t> head, gh; rbuf<uint8_t> pend;
		if (!pos.alloc(m) || !key.alloc(m) || !ps.alloc(m) || !ks.alloc(m) || !head.alloc(m) || !gh.alloc(m) || !pend.alloc(m)) { ssg_err_msg = "index construction: device allocation failed (bucket)"; rc = SSG_ENOMEM; break; }
		SSG_LAUNCH(ssg_k_idx_bucket_scatter, n_wg, 256, 0, d_T, n, p, b, wg_off.p, pos.p);
		SSG_LAUNCH(ssg_k_idx_key, nblk256((int64_t)m), 256, 0, d_T, pos.p, (int64_t)m, p, key.p);
		if ((rc = prim_sort_pairs_u64(key.p, ks.p, pos.p, ps.p, (int64_t)m, 0, 64))) break;
		SSG_LAUNCH(ssg_k_idx_heads, nblk256((int64_t)m), 256, 0, ks.p, (int64_t)m, head.p);
		if ((rc = prim_scan_max_i64(head.p, gh.p, (int64_t)m))) break;
		grp.p = key.take(); grp.n = m;   /* the unsorted keys are dead: reuse */
		SSG_LAUNCH(ssg_k_idx_place, nblk256((int64_t)m), 256, 0, ps.p, gh.p, (int64_t)m, base, d_SA, rank.p, grp.p, pend.p);
		uint64_t np = 0;
		if ((rc = prim_count_flags(pend.p, (int64_t)m, &np))) break;
		if (np) {
			chunk_t c; c.m = np;
			c.pos = (uint64_t*)rt_malloc_raw(np * 8); c.grp = (uint64_t*)rt_malloc_raw(np * 8);
			chunks.push_back(c);
			if (!c.pos || !c.grp) { ssg_err_msg = "index construction: device allocation failed (pending)"; rc = SSG_ENOMEM; break; }
			uint64_t g1 = 0, g2 = 0;
			if ((rc = prim_select_u64(ps.p, 0, pend.p, (int64_t)m, c.pos, &g1))) break;
			if ((rc = prim_select_u64(grp.p, 0, pend.p, (int64_t)m, c.grp, &g2))) break;
			n_pend += np;
		}
		if (idx_verbose()) fprintf(stderr, "[ssg index] bucket %u/%u: %llu suffixes, %llu not yet unique\n", b, n_bucket, (unsigned long long)m, (unsigned long long)np);
		base += m;
		if ((rc = rt_sync())) break;
	}
	if (!rc && base != (uint64_t)n) { ssg_err_msg = "index construction: buckets do not cover the text"; rc = SSG_EHIP; }
	if (rc) { free_chunks(); return rc; }
	wg_cnt.release(); wg_off.release();
	/* ---- pending suffixes (SA order) ---- */
	rbuf<uint64_t> P_pos, P_grp;
	if (n_pend) {
		if (!P_pos.alloc(n_pend) || !P_grp.alloc(n_pend)) { free_chunks(); ssg_err_msg = "index construction: device allocation failed (pending list)"; return SSG_ENOMEM; }
		uint64_t o = 0;
		for (auto &c : chunks) { rc |= rt_d2d(P_pos.p + o, c.pos, c.m * 8); rc |= rt_d2d(P_grp.p + o, c.grp, c.m * 8); o += c.m; }
	}
	free_chunks();
	if (rc) return SSG_EHIP;
	/* ---- prefix doubling on the pending suffixes only ---- */
	uint64_t h = (uint64_t)p + SSG_IDX_KEYSYM;
	int bits_n = 1; while (bits_n < 64 && ((2 * (uint64_t)n) >> bits_n)) ++bits_n;   /* key2 < 2n, group index < n */
	for (int round = 2; n_pend; ++round, h <<= 1) {
		const int64_t m = (int64_t)n_pend;
		if (idx_verbose()) fprintf(stderr, "[ssg index] round %d: h = %llu, %lld suffixes pending\n", round, (unsigned long long)h, (long long)m);
		rbuf<uint64_t> key2, iota, k2a, perma, grpa, grps, permb, poss, k2s, saidx, grpn; rbuf<int64_t> oh, ohs, nh, nhs; rbuf<uint8_t> pend;
		RALLOC(key2, m); RALLOC(iota, m); RALLOC(k2a, m); RALLOC(perma, m);
		SSG_LAUNCH(ssg_k_idx_key2, nblk256(m), 256, 0, P_pos.p, m, h, (uint64_t)n, rank.p, key2.p, iota.p);
		CHK(prim_sort_pairs_u64(key2.p, k2a.p, iota.p, perma.p, m, 0, bits_n));
		iota.release(); k2a.release();
		RALLOC(grpa, m); RALLOC(grps, m); RALLOC(permb, m);
		SSG_LAUNCH(ssg_k_idx_gather, nblk256(m), 256, 0, perma.p, P_grp.p, m, grpa.p);
		CHK(prim_sort_pairs_u64(grpa.p, grps.p, perma.p, permb.p, m, 0, bits_n));
		grpa.release(); perma.release();
		RALLOC(poss, m); RALLOC(k2s, m);
		SSG_LAUNCH(ssg_k_idx_gather, nblk256(m), 256, 0, permb.p, P_pos.p, m, poss.p);
		SSG_LAUNCH(ssg_k_idx_gather, nblk256(m), 256, 0, permb.p, key2.p, m, k2s.p);
		CHK(rt_sync());
		permb.release(); key2.release();
		RALLOC(oh, m); RALLOC(ohs, m);
		SSG_LAUNCH(ssg_k_idx_oldheads, nblk256(m), 256, 0, grps.p, m, oh.p);
		CHK(prim_scan_max_i64(oh.p, ohs.p, m));
		oh.release();
		RALLOC(saidx, m); RALLOC(nh, m); RALLOC(nhs, m);
		SSG_LAUNCH(ssg_k_idx_newheads, nblk256(m), 256, 0, grps.p, k2s.p, ohs.p, m, saidx.p, nh.p);
		CHK(prim_scan_max_i64(nh.p, nhs.p, m));
		nh.release(); ohs.release(); grps.release(); k2s.release();
		RALLOC(grpn, m); RALLOC(pend, m);
		SSG_LAUNCH(ssg_k_idx_replace, nblk256(m), 256, 0, poss.p, saidx.p, nhs.p, m, d_SA, rank.p, grpn.p, pend.p);
		uint64_t np = 0;
		CHK(prim_count_flags(pend.p, m, &np));
		saidx.release(); nhs.release();
		rbuf<uint64_t> npos, ngrp;
		if (np) {
			uint64_t g1, g2;
			RALLOC(npos, np); RALLOC(ngrp, np);
			CHK(prim_select_u64(poss.p, 0, pend.p, m, npos.p, &g1));
			CHK(prim_select_u64(grpn.p, 0, pend.p, m, ngrp.p, &g2));
		}
		CHK(rt_sync());
		P_pos.release(); P_grp.release();
		P_pos.p = npos.take(); P_grp.p = ngrp.take();
		n_pend = np;
		if (h > (uint64_t)n) { ssg_err_msg = "index construction: prefix doubling did not converge"; return SSG_EHIP; }
	}
	return 0;
}

/* everything the aligner needs, derived from the forward strand resident in HBM */
static int build_from_fwd(const uint8_t *d_fwd, int64_t l_pac, ssg_index *ix)
{
	if (l_pac <= 0) { ssg_err_msg = "index construction: empty reference"; return SSG_EINVAL; }
	const int64_t n = 2 * l_pac;
	int sa_intv = 4;
	{ const char *e = getenv("SSG_SA_INTV"); if (e && *e) sa_intv = atoi(e); }
	if (sa_intv <= 0 || sa_intv > 32 || (sa_intv & (sa_intv - 1))) { ssg_err_msg = "SSG_SA_INTV must be a power of two <= 32"; return SSG_EINVAL; }
	rbuf<uint8_t> T; rbuf<uint64_t> SA;
	RALLOC(T, (size_t)n + 256); RALLOC(SA, (size_t)n);
	SSG_LAUNCH(ssg_k_idx_text, nblk256(n + 256), 256, 0, d_fwd, l_pac, T.p, n + 256);
	CHK(build_suffix_array(T.p, n, SA.p));
	/* primary = with-$ row of suffix 0; find it with a one-flag select over the suffix array */
	uint64_t primary = 0;
	{	/* SA[r-1] == 0  <=>  row r is the primary row */
		rbuf<uint64_t> one;
		RALLOC(one, 1);
		SSG_LAUNCH(ssg_k_idx_find_primary, nblk256(n), 256, 0, SA.p, (uint64_t)n, one.p);
		CHK(rt_sync());
		CHK(rt_d2h(&primary, one.p, 8));
	}
	if (primary == 0 || primary > (uint64_t)n) { ssg_err_msg = "index construction: primary row not found"; return SSG_EHIP; }
	/* sampled SA for the HBM-resident index */
	const uint64_t n_sa = ((uint64_t)n + sa_intv) / sa_intv;
	rbuf<uint64_t> samp;
	RALLOC(samp, n_sa);
	SSG_LAUNCH(ssg_k_idx_sa_sample, nblk256((int64_t)n_sa), 256, 0, SA.p, n_sa, sa_intv, samp.p);
	/* BWT symbols, then the interleaved body */
	rbuf<uint8_t> B;
	RALLOC(B, (size_t)n);
	SSG_LAUNCH(ssg_k_idx_bwt_sym, nblk256(n), 256, 0, T.p, SA.p, (uint64_t)n, primary, B.p);
	CHK(rt_sync());
	SA.release(); T.release();
	const uint64_t nblk = ((uint64_t)n + 127) / 128, n_words = ((uint64_t)n + 15) / 16, bwt_words = n_words + 8 * (nblk + 1);
	rbuf<uint32_t> words, cnt, bwt; rbuf<uint64_t> occ;
	RALLOC(words, nblk * 8); RALLOC(cnt, 4 * nblk + 4); RALLOC(occ, 4 * (nblk + 1)); RALLOC(bwt, bwt_words + 64);
	CHK(rt_memset(cnt.p, 0, (4 * nblk + 4) * 4)); CHK(rt_memset(bwt.p, 0, (bwt_words + 64) * 4));
	SSG_LAUNCH(ssg_k_idx_bwt_pack, nblk256((int64_t)nblk), 256, 0, B.p, (uint64_t)n, nblk, words.p, cnt.p);
	CHK(rt_sync());
	B.release();
	{	/* per-symbol exclusive sums; each scan reads one element past its count row (the next row's first count or the spare
		 * slot), which only feeds out[nblk + 1] -- never written because exactly nblk + 1 outputs are produced */
		for (int c = 0; c < 4; ++c) {
			rbuf<uint32_t> row;   /* a private zero-terminated copy keeps the scan's n + 1 reads inside the row */
			RALLOC(row, nblk + 1);
			CHK(rt_memset(row.p, 0, (nblk + 1) * 4));
			CHK(rt_d2d(row.p, cnt.p + (size_t)c * nblk, nblk * 4));
			CHK(prim_exsum_u32_u64(row.p, occ.p + (size_t)c * (nblk + 1), (int64_t)nblk));
		}
	}
	SSG_LAUNCH(ssg_k_idx_bwt_write, nblk256((int64_t)nblk + 1), 256, 0, words.p, occ.p, (uint64_t)n, nblk, bwt.p);
	uint64_t tot[4];
	for (int c = 0; c < 4; ++c) CHK(rt_d2h(&tot[c], occ.p + (size_t)c * (nblk + 1) + nblk, 8));
	const size_t pac_bytes = (size_t)(l_pac / 4 + 1);
	rbuf<uint8_t> pac;
	RALLOC(pac, pac_bytes + 64);
	SSG_LAUNCH(ssg_k_idx_pac, nblk256((int64_t)pac_bytes), 256, 0, d_fwd, l_pac, pac.p, (int64_t)pac_bytes);
	CHK(rt_sync());
	ix->v.primary = primary; ix->v.L2[0] = 0;
	for (int c = 0; c < 4; ++c) ix->v.L2[c + 1] = ix->v.L2[c] + tot[c];
	if (ix->v.L2[4] != (uint64_t)n) { ssg_err_msg = "index construction: symbol counts do not add up"; return SSG_EHIP; }
	ix->v.seq_len = (uint64_t)n; ix->v.l_pac = l_pac; ix->v.sa_intv = sa_intv;
	ix->bwt_words = bwt_words; ix->raw_alloc = true;
	ix->bwt = bwt.take(); ix->sa = samp.take(); ix->pac = pac.take();
	ix->v.bwt = ix->bwt; ix->v.sa = ix->sa; ix->v.pac = ix->pac;
	rt_pool_release();   /* the primitives' temporaries went through the arena */
	return 0;
}

static int set_contigs(ssg_index *ix, int n_ctg, const int64_t *ctg_off, const int32_t *ctg_len)
{
	ix->ctg_off = (int64_t*)rt_malloc((size_t)n_ctg * 8); ix->ctg_len = (int32_t*)rt_malloc((size_t)n_ctg * 4);
	if (!ix->ctg_off || !ix->ctg_len) { ssg_err_msg = "index allocation failed"; return SSG_ENOMEM; }
	CHK(rt_h2d(ix->ctg_off, ctg_off, (size_t)n_ctg * 8)); CHK(rt_h2d(ix->ctg_len, ctg_len, (size_t)n_ctg * 4));
	ix->v.ctg_off = ix->ctg_off; ix->v.ctg_len = ix->ctg_len; ix->v.n_ctg = n_ctg;
	ix->h_off.assign(ctg_off, ctg_off + n_ctg); ix->h_len.assign(ctg_len, ctg_len + n_ctg);
	return 0;
}

SSG_ABI_FP_DEFINE(index_build)
extern "C" {

int ssg_index_build_dev(const uint8_t *d_fwd, int64_t l_pac, int n_ctg, const int64_t *ctg_off, const int32_t *ctg_len, ssg_index_t **out)
{
	*out = 0;
	if (rt_device_count() < 1) { ssg_err_msg = "no HIP device visible: libssgpu has no CPU path"; return SSG_ENODEV; }
	{ const int rc0 = ssg_abi_selfcheck(); if (rc0) return rc0; }
	if (n_ctg < 1) { ssg_err_msg = "ssg_index_build_dev: no contigs"; return SSG_EINVAL; }
	ssg_index *ix = new ssg_index();
	int rc = build_from_fwd(d_fwd, l_pac, ix);
	if (!rc) rc = set_contigs(ix, n_ctg, ctg_off, ctg_len);
	if (!rc) rc = ssg_index_build_ktab(ix);
	if (rc) { ssg_index_destroy(ix); return rc; }
	ix->names.resize(n_ctg); ix->annos.assign(n_ctg, ""); ix->n_ambs.assign(n_ctg, 0);
	for (int i = 0; i < n_ctg; ++i) ix->names[i] = std::to_string(i + 1);
	*out = ix;
	return 0;
}

/* upstream bns_fasta2bntseq (bntseq.c): names/comments with kseq semantics, every non-ACGT base becomes
 * lrand48() & 3 (srand48(11)) and is recorded as a hole; runs of the same ambiguity code form one hole */
int ssg_index_build_fasta(const char *fasta, ssg_index_t **out)
{
	*out = 0;
	if (rt_device_count() < 1) { ssg_err_msg = "no HIP device visible: libssgpu has no CPU path"; return SSG_ENODEV; }
	{ const int rc0 = ssg_abi_selfcheck(); if (rc0) return rc0; }
	gzFile fp = gzopen(fasta, "r");
	if (!fp) { ssg_err_msg = std::string("cannot open ") + fasta; return SSG_EIO; }
	gzbuffer(fp, 1 << 20);
	ssg_index *ix = new ssg_index();
	std::vector<uint8_t> codes; std::vector<int64_t> off; std::vector<int32_t> len;
	uint8_t lut[256]; memset(lut, 4, 256);
	lut['A'] = lut['a'] = 0; lut['C'] = lut['c'] = 1; lut['G'] = lut['g'] = 2; lut['T'] = lut['t'] = 3;
	srand48(11);
	std::vector<char> buf(1 << 20);
	int lasts = 0; bool in_hdr = false, at_bol = true, bad = false; std::string hdr; int64_t cur_len = 0;
	auto end_seq = [&]() { if (!off.empty()) { if (cur_len > 0x7fffffffLL) bad = true; len.push_back((int32_t)cur_len); } };
	auto begin_seq = [&]() {
		size_t e = 0; while (e < hdr.size() && !isspace((unsigned char)hdr[e])) ++e;
		ix->names.push_back(hdr.substr(0, e));
		/* kseq_read: the name ends at the first blank, which alone is consumed; the comment is the rest of the line (further blanks
		 * included), minus one trailing CR when it is longer than one character (ks_getuntil2, kseq.h:143) */
		std::string c = e < hdr.size() ? hdr.substr(e + 1) : std::string();
		if (c.size() > 1 && c.back() == '\r') c.pop_back();
		ix->annos.push_back(c); ix->n_ambs.push_back(0);
		off.push_back((int64_t)codes.size()); cur_len = 0; lasts = 0;
	};
	for (;;) {
		int got = gzread(fp, buf.data(), (unsigned)buf.size());
		if (got <= 0) break;
		for (int i = 0; i < got; ++i) {
			const unsigned char ch = (unsigned char)buf[i];
			if (in_hdr) { if (ch == '\n') { in_hdr = false; at_bol = true; begin_seq(); } else hdr += (char)ch; continue; }
			if (ch == '\n') { at_bol = true; continue; }
			if (at_bol && ch == '>') { end_seq(); in_hdr = true; hdr.clear(); continue; }
			at_bol = false;
			if (isspace(ch) || off.empty()) continue;          /* kseq keeps graphic characters only */
			int c = lut[ch];
			if (c >= 4) {
				if (lasts == ch && !ix->holes.empty()) ++ix->holes.back().len;
				else { ssg_hole_t q; q.offset = (int64_t)codes.size(); q.len = 1; q.amb = (char)ch; ix->holes.push_back(q); ++ix->n_ambs.back(); }
				lasts = ch;
				c = (int)(lrand48() & 3);
			} else lasts = 0;
			codes.push_back((uint8_t)c); ++cur_len;
		}
	}
	if (in_hdr) begin_seq();
	end_seq();
	gzclose(fp);
	if (bad || off.empty() || codes.empty()) { delete ix; ssg_err_msg = "no sequence in the FASTA file (or a contig longer than 2^31)"; return SSG_EINVAL; }
	const int64_t l_pac = (int64_t)codes.size();
	rbuf<uint8_t> d_fwd;
	if (!d_fwd.alloc((size_t)l_pac)) { delete ix; ssg_err_msg = "index construction: device allocation failed (reference)"; return SSG_ENOMEM; }
	int rc = rt_h2d(d_fwd.p, codes.data(), (size_t)l_pac);
	{ std::vector<uint8_t>().swap(codes); }
	if (!rc) rc = build_from_fwd(d_fwd.p, l_pac, ix);
	if (!rc) rc = set_contigs(ix, (int)off.size(), off.data(), len.data());
	if (!rc) rc = ssg_index_build_ktab(ix);
	if (rc) { ssg_index_destroy(ix); return rc; }
	*out = ix;
	return 0;
}

/* writes prefix.{amb,ann,pac,bwt,sa} exactly as upstream bwa index leaves them (bns_dump, bwt_dump_bwt, bwt_dump_sa; .sa interval 32) */
int ssg_index_save(const ssg_index_t *ix, const char *prefix)
{
	if (!ix->bwt || !ix->bwt_words) { ssg_err_msg = "ssg_index_save: the index does not own its arrays"; return SSG_EINVAL; }
	const std::string p(prefix);
	const int n_ctg = ix->v.n_ctg; const int64_t l_pac = ix->v.l_pac;
	FILE *fp = fopen((p + ".ann").c_str(), "w");
	if (!fp) { ssg_err_msg = "cannot write " + p + ".ann"; return SSG_EIO; }
	fprintf(fp, "%lld %d %u\n", (long long)l_pac, n_ctg, 11u);
	for (int i = 0; i < n_ctg; ++i) {
		const std::string &an = i < (int)ix->annos.size() ? ix->annos[i] : std::string();
		fprintf(fp, "%d %s", 0, i < (int)ix->names.size() ? ix->names[i].c_str() : "*");
		if (!an.empty()) fprintf(fp, " %s\n", an.c_str()); else fprintf(fp, " (null)\n");
		fprintf(fp, "%lld %d %d\n", (long long)ix->h_off[i], ix->h_len[i], i < (int)ix->n_ambs.size() ? ix->n_ambs[i] : 0);
	}
	fclose(fp);
	fp = fopen((p + ".amb").c_str(), "w");
	if (!fp) { ssg_err_msg = "cannot write " + p + ".amb"; return SSG_EIO; }
	fprintf(fp, "%lld %d %u\n", (long long)l_pac, n_ctg, (unsigned)ix->holes.size());
	for (const ssg_hole_t &q : ix->holes) fprintf(fp, "%lld %d %c\n", (long long)q.offset, q.len, q.amb);
	fclose(fp);
	auto dump = [&](const std::string &fn, const void *hdr, size_t hdr_bytes, const void *dev, size_t bytes, const void *tail, size_t tail_bytes) -> int {
		FILE *f = fopen(fn.c_str(), "wb");
		if (!f) { ssg_err_msg = "cannot write " + fn; return SSG_EIO; }
		int rc = 0;
		if (hdr_bytes && fwrite(hdr, 1, hdr_bytes, f) != hdr_bytes) rc = SSG_EIO;
		const size_t CH = (size_t)256 << 20;
		std::vector<uint8_t> h(std::min(bytes, CH) + 1);
		for (size_t o = 0; o < bytes && !rc; o += CH) {
			const size_t k = std::min(CH, bytes - o);
			rc = rt_d2h(h.data(), (const uint8_t*)dev + o, k);
			if (!rc && fwrite(h.data(), 1, k, f) != k) rc = SSG_EIO;
		}
		if (!rc && tail_bytes && fwrite(tail, 1, tail_bytes, f) != tail_bytes) rc = SSG_EIO;
		if (fclose(f) != 0) rc = SSG_EIO;
		if (rc == SSG_EIO) ssg_err_msg = "short write on " + fn;
		return rc;
	};
	{	/* .pac: ceil(l_pac / 4) bytes, an extra zero byte when l_pac % 4 == 0, then l_pac % 4 */
		uint8_t tail[2]; size_t nt = 0;
		if (l_pac % 4 == 0) tail[nt++] = 0;
		tail[nt++] = (uint8_t)(l_pac % 4);
		CHK(dump(p + ".pac", 0, 0, ix->pac, (size_t)(l_pac / 4 + (l_pac % 4 ? 1 : 0)), tail, nt));
	}
	uint64_t hdr[7] = { ix->v.primary, ix->v.L2[1], ix->v.L2[2], ix->v.L2[3], ix->v.L2[4], 32, ix->v.seq_len };
	CHK(dump(p + ".bwt", hdr, 40, ix->bwt, (size_t)ix->bwt_words * 4, 0, 0));
	{	/* .sa at upstream's interval 32: every (32 / sa_intv)-th sample of the HBM-resident array, without row 0 */
		const int stride = 32 / ix->v.sa_intv;
		if (stride < 1) { ssg_err_msg = "ssg_index_save: resident SA is sparser than 32"; return SSG_EINVAL; }
		const uint64_t n32 = (ix->v.seq_len + 32) / 32;
		rbuf<uint64_t> s32;
		RALLOC(s32, n32);
		SSG_LAUNCH(ssg_k_idx_stride_u64, nblk256((int64_t)n32), 256, 0, ix->sa, n32, stride, s32.p);
		CHK(rt_sync());
		CHK(dump(p + ".sa", hdr, 56, s32.p + 1, (size_t)(n32 - 1) * 8, 0, 0));
	}
	return 0;
}

} /* extern "C" */
