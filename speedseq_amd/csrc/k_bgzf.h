/*
 * k_bgzf.h -- BGZF deflate on the device (the format: htslib bgzf.c:298-342 -- independent gzip members of
 * at most 0xff00 payload bytes, RFC 1951 deflate inside).  The coordinate sort's last step writes every record once more, compressed, and
 * on a host whose cores are busy (or few) that step was half of the wall time of `speedseq align` behind an MI355X; the payload of a block
 * is independent of every other block, i.e. a wave's worth of work each.
 *
 * One wave per block:
 *   1. LZ77: the block goes by in regions of 4 KB, lane l parses the l-th 64 bytes of the region (greedy with one step of lazy matching) --
 *      hash of the next 4 bytes into two tables SHARED by the wave (LDS): the region's own positions, entered before the parse (1024 buckets
 *      of four; a lane takes what lies before its position), and everything before the region (2048 buckets of two); plus the byte just
 *      behind (runs).  A match lies <= 32768 back and ends inside the lane's 64 bytes.  Symbols (literal / length + distance) go to a scratch list per
 *      lane, their frequencies to LDS counters.
 *   2. Huffman code lengths of the literal/length and distance alphabets and of the code-length alphabet (RFC 1951 3.2.7), by one lane:
 *      leaves sorted by frequency, two-queue merge; a tree deeper than the format allows (15 / 7 bits) is rebuilt on halved frequencies
 *      (always terminates, always a complete code -- zlib's inflate rejects incomplete ones).  At least two codes per alphabet, as zlib.
 *   3. every lane sizes its symbols, a prefix sum places them, and the lanes write their bits side by side (LSB-first; full words by plain
 *      stores, the word shared with a neighbour by atomic OR into the zeroed output).  A block that does not shrink is stored.
 * Output per block: the raw deflate stream (the caller adds the 18-byte BGZF header, CRC-32 and ISIZE).  Not byte-identical to zlib's
 * output -- no two deflate implementations are -- but any inflate gives back the payload; tests inflate every block with zlib.
 */
#ifndef SSG_K_BGZF_H
#define SSG_K_BGZF_H
#include "ssg_dev.h"

#define BZ_MAX_PAYLOAD 0xff00     /* htslib BGZF_BLOCK_SIZE */
#define BZ_HBITS 12
#define BZ_CBITS 12              /* the region's own table: 1024 buckets of the four newest positions */
#define BZ_CHUNK 64               /* bytes a lane parses per region; a match ends inside its chunk */
#define BZ_MAX_REGIONS 16         /* ceil(0xff00 / (64 * BZ_CHUNK)) */
#define BZ_STRETCH_CAP 1024       /* symbols per lane: at most BZ_MAX_REGIONS * BZ_CHUNK */
#define BZ_OUT_STRIDE 65536       /* bytes of temporary output per block: the stored form is payload + 5 */

SSG_DEVFN uint32_t bz_load32(const uint8_t *p) { uint32_t w; memcpy(&w, p, 4); return w; }
SSG_DEVFN int bz_log2(uint32_t v) { return 31 - __builtin_clz(v); }
/* RFC 1951 3.2.5: length 3..258 -> code 0..28 (symbol 257 + code), extra bits */
SSG_DEVFN void bz_len_code(int len, int &code, int &ebits, int &eval)
{
	const int l = len - 3;
	if (l < 8) { code = l; ebits = 0; eval = 0; }
	else if (len == 258) { code = 28; ebits = 0; eval = 0; }
	else { const int e = bz_log2((uint32_t)l) - 2; code = 4 * e + 4 + ((l >> e) & 3); ebits = e; eval = l & ((1 << e) - 1); }
}
/* distance 1..32768 -> code 0..29, extra bits */
SSG_DEVFN void bz_dist_code(int dist, int &code, int &ebits, int &eval)
{
	const int d = dist - 1;
	if (d < 4) { code = d; ebits = 0; eval = 0; }
	else { const int e = bz_log2((uint32_t)d) - 1; code = 2 * e + 2 + ((d >> e) & 1); ebits = e; eval = d & ((1 << e) - 1); }
}
SSG_DEVFN uint32_t bz_rev(uint32_t c, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) { r = r << 1 | (c & 1); c >>= 1; } return r; }

/* Huffman code lengths (<= maxbits) of n symbols with frequencies f[] (one lane).  Work arrays: idx[n], wt[2n], par[2n] (int32, n <= 288).
 * Symbols of frequency 0 get length 0; fewer than two used symbols are topped up with symbols 0 / 1 (zlib build_tree does the same: every
 * alphabet is sent with at least two codes, so that its code is complete). */
SSG_DEVFN void bz_code_lengths(uint32_t *f, int n, int maxbits, uint8_t *len, int32_t *idx, uint32_t *wt, int32_t *par)
{
	int m = 0;
	for (int s = 0; s < n; ++s) { len[s] = 0; if (f[s]) ++m; }
	for (int s = 0; m < 2 && s < n; ++s) if (!f[s]) { f[s] = 1; ++m; }
	for (;;) {
		m = 0;
		for (int s = 0; s < n; ++s) if (f[s]) {   /* insertion sort by (frequency, symbol) */
			int k = m++;
			while (k > 0 && f[idx[k - 1]] > f[s]) { idx[k] = idx[k - 1]; --k; }
			idx[k] = s;
		}
		for (int k = 0; k < m; ++k) wt[k] = f[idx[k]];
		int leaf = 0, inode = m, made = m;   /* two queues: leaves [leaf, m), internal nodes [inode, made) -- both in non-decreasing weight */
		while (made < 2 * m - 1) {
			int a, b;
			if (leaf < m && (inode >= made || wt[leaf] <= wt[inode])) a = leaf++; else a = inode++;
			if (leaf < m && (inode >= made || wt[leaf] <= wt[inode])) b = leaf++; else b = inode++;
			wt[made] = wt[a] + wt[b]; par[a] = made; par[b] = made; ++made;
		}
		par[2 * m - 2] = -1;
		int deepest = 0;
		/* depth of a node = depth of its parent + 1; parents have larger indices: walk down from the root, depths kept in wt[] (weights are done with) */
		wt[2 * m - 2] = 0;
		for (int k = 2 * m - 3; k >= 0; --k) { wt[k] = wt[par[k]] + 1; if (k < m && (int)wt[k] > deepest) deepest = (int)wt[k]; }
		if (deepest <= maxbits) { for (int k = 0; k < m; ++k) len[idx[k]] = (uint8_t)wt[k]; return; }
		for (int s = 0; s < n; ++s) if (f[s]) f[s] = (f[s] + 1) >> 1;   /* flatter frequencies, shallower tree */
	}
}
/* canonical codes of the lengths (RFC 1951 3.2.2), bit-reversed for the LSB-first stream */
SSG_DEVFN void bz_codes(const uint8_t *len, int n, int maxbits, uint16_t *code)
{
	int cnt[16]; for (int b = 0; b < 16; ++b) cnt[b] = 0;
	for (int s = 0; s < n; ++s) ++cnt[len[s]];
	cnt[0] = 0;
	uint32_t next[16]; uint32_t c = 0;
	for (int b = 1; b <= maxbits; ++b) { c = (c + (uint32_t)cnt[b - 1]) << 1; next[b] = c; }
	for (int s = 0; s < n; ++s) code[s] = len[s] ? (uint16_t)bz_rev(next[len[s]]++, len[s]) : 0;
}

/* LSB-first bit writer of one lane into the block's zeroed output: the first and the last word of the lane's range are shared with the
 * neighbouring lanes (atomic OR), the words between are the lane's alone */
struct bz_writer_t { uint32_t *out; uint64_t acc; int nacc; uint32_t word; bool shared; };
SSG_DEVFN void bz_w_init(bz_writer_t &w, uint32_t *out, uint64_t bit0) { w.out = out; w.word = (uint32_t)(bit0 >> 5); w.nacc = (int)(bit0 & 31); w.acc = 0; w.shared = w.nacc != 0; }
SSG_DEVFN void bz_w_put(bz_writer_t &w, uint32_t bits, int n)
{
	w.acc |= (uint64_t)bits << w.nacc; w.nacc += n;
	if (w.nacc >= 32) {
		const uint32_t v = (uint32_t)w.acc;
		if (w.shared) { atomicOr(w.out + w.word, v); w.shared = false; } else w.out[w.word] = v;
		++w.word; w.acc >>= 32; w.nacc -= 32;
	}
}
SSG_DEVFN void bz_w_end(bz_writer_t &w) { if (w.nacc > 0) atomicOr(w.out + w.word, (uint32_t)w.acc); }

/*
 * payload: the concatenated block payloads, cut[b] .. cut[b+1] delimits block blk0 + b (<= BZ_MAX_PAYLOAD bytes).  tmp: BZ_OUT_STRIDE bytes
 * per launched wave... per block of this launch; sym: BZ_STRETCH_CAP * 64 words per block.  size[b] = bytes of the block's deflate stream in tmp.
 */
__global__ void __launch_bounds__(64) ssg_k_bgzf_deflate(const uint8_t *payload, const uint64_t *cut, int n_blocks, uint8_t *tmp, uint32_t *sym, uint32_t *size)
{
	/* the two hash tables are dead when the codes are built: the work arrays of the code builder (5.8 KB) lie over them, and a CU holds ten waves of this kernel instead of seven */
	__shared__ uint32_t lz_pool[((1 << BZ_HBITS) + (1 << BZ_CBITS)) / 2];
	uint16_t *const ht = (uint16_t*)lz_pool, *const hc = ht + (1 << BZ_HBITS);
	static_assert(sizeof(lz_pool) >= (288 + 576 + 576) * 4, "the code builder's work arrays fit the hash tables' space");
	int32_t *const w_idx = (int32_t*)lz_pool; uint32_t *const w_wt = lz_pool + 288; int32_t *const w_par = (int32_t*)lz_pool + 288 + 576;
	__shared__ uint32_t f_ll[288], f_d[32], f_cl[19];
	__shared__ uint8_t l_ll[288], l_d[32], l_cl[19];
	__shared__ uint16_t c_ll[288], c_d[32], c_cl[19];
	__shared__ uint16_t rle[320];            /* the code-length sequence: symbol | extra value << 5 */
	__shared__ uint32_t sh_misc[8];          /* 0: n_rle, 1: hlit, 2: hdist, 3: hclen, 4: header bits */
	const int b = (int)blockIdx.x, lane = wv_lane();
	if (b >= n_blocks) return;
	const uint8_t *src = payload + cut[b];
	const int n = (int)(cut[b + 1] - cut[b]);
	uint32_t *const out = (uint32_t*)(tmp + (size_t)b * BZ_OUT_STRIDE);
	uint32_t *const my_sym = sym + (size_t)b * BZ_STRETCH_CAP * 64 + lane;
	for (int k = lane; k < (1 << BZ_HBITS) / 2; k += 64) ((uint32_t*)ht)[k] = 0xffffffffu;
	for (int k = lane; k < (1 << BZ_CBITS) / 2; k += 64) ((uint32_t*)hc)[k] = 0xffffffffu;
	for (int k = lane; k < 288; k += 64) f_ll[k] = 0;
	if (lane < 32) f_d[lane] = 0;
	if (lane < 19) f_cl[lane] = 0;
	for (int k = lane; k < BZ_OUT_STRIDE / 4; k += 64) out[k] = 0;
	ssg_wave_ldssync();
	/* ---- 1. LZ77: the block goes by in regions of 64 x BZ_CHUNK bytes, lane l parses chunk l of the region.  The lanes run in lock step, so a table
	 * filled while parsing shows a lane only the parts of the other chunks at smaller offsets than its own -- half of the previous record, in a
	 * BAM.  Hence two tables: the region's positions all enter `hc' BEFORE the parse (a lane uses those before its position), and `ht' holds
	 * what came before the region.  Sizes on a sorted BAM stream (zlib level 6 = 0.204, level 1 = 0.224 of the payload): one stretch per lane
	 * and one table filled while parsing 0.29 (and 0.61 vs 0.47 on the synthetic records of the tests); this form 0.232. ---- */
	const int n_regions = (n + 64 * BZ_CHUNK - 1) / (64 * BZ_CHUNK);
	int ns = 0, s1 = 0;
	uint16_t chunk_syms[BZ_MAX_REGIONS];
	auto match_len = [&](const int cand, const int p, const int maxl) -> int {
		int l = 0;
		while (l + 4 <= maxl && bz_load32(src + cand + l) == bz_load32(src + p + l)) l += 4;
		while (l < maxl && src[cand + l] == src[p + l]) ++l;
		return l;
	};
	/* the longest match at position p among the two table entries of its hash and the position just before it */
	auto find = [&](const int p, int &mlen, int &mdist) {
		mlen = 0; mdist = 0;
		const int maxl = s1 - p < 258 ? s1 - p : 258;
		if (p > 0 && maxl >= 4) { const int l = match_len(p - 1, p, maxl); if (l >= 4) { mlen = l; mdist = 1; } }
		if (p + 4 > n) return;
		const uint32_t wp = bz_load32(src + p), w4 = wp * 2654435761u, h = (w4 >> (32 - BZ_HBITS)) & ~1u, g = (w4 >> (32 - BZ_CBITS)) & ~3u;
		/* the six candidates' first four bytes are fetched together (one round trip instead of six one after the other: the parse is a chain of dependent loads, and a
		 * candidate that differs there -- most do: the tables keep positions by hash -- could not give the four bytes a match needs); the choice is as before */
		int cnd[6]; uint32_t wc[6];
		SSG_UNROLL for (int t = 0; t < 6; ++t) {
			cnd[t] = t < 4 ? (int)hc[g + t] : (int)ht[h + t - 4];
			const bool ok = cnd[t] != 0xffff && cnd[t] < p && p - cnd[t] <= 32768;
			wc[t] = ok ? bz_load32(src + cnd[t]) : ~wp;
		}
		SSG_UNROLL for (int t = 0; t < 6; ++t) {   /* the region's own positions first (nearer: cheaper distances), then the two newest from before it */
			if (wc[t] == wp && maxl >= 4 && !(t >= 4 && mlen >= 16)) {   /* a good match nearby: the older table is not asked */
				const int l = match_len(cnd[t], p, maxl);
				if (l >= 4 && l > mlen) { mlen = l; mdist = p - cnd[t]; }
			}
		}
	};
	auto literal = [&](const int p) { const uint32_t c = src[p]; atomicAdd(&f_ll[c], 1u); my_sym[(size_t)ns * 64] = c; ++ns; };
	SSG_UNROLL for (int j = 0; j < BZ_MAX_REGIONS; ++j) {
		const int ns0 = ns;
		if (j < n_regions) {
			const int c0 = (j * 64 + lane) * BZ_CHUNK, s0 = c0 < n ? c0 : n;
			s1 = s0 + BZ_CHUNK < n ? s0 + BZ_CHUNK : n;
			for (int q = s0; q < s1 && q + 4 <= n; ++q) {   /* the region's positions into its own table first: a lane takes from it what lies before its position */
				const uint32_t gq = ((bz_load32(src + q) * 2654435761u) >> (32 - BZ_CBITS)) & ~3u;
				hc[gq + 3] = hc[gq + 2]; hc[gq + 2] = hc[gq + 1]; hc[gq + 1] = hc[gq]; hc[gq] = (uint16_t)q;
			}
			ssg_wave_ldssync();
			for (int pos = s0; pos < s1; ) {
				int mlen, mdist;
				find(pos, mlen, mdist);
				if (mlen && mlen < 32 && pos + 1 < s1) {   /* one step of lazy matching: a longer match one byte on is worth a literal */
					int l2, d2;
					find(pos + 1, l2, d2);
					if (l2 > mlen + 1) { literal(pos); ++pos; mlen = l2; mdist = d2; }
				}
				if (mlen) {
					int lc, le, lv, dc, de, dv;
					bz_len_code(mlen, lc, le, lv); bz_dist_code(mdist, dc, de, dv);
					atomicAdd(&f_ll[257 + lc], 1u); atomicAdd(&f_d[dc], 1u);
					my_sym[(size_t)ns * 64] = 0x80000000u | (uint32_t)mlen << 16 | (uint32_t)(mdist - 1); ++ns;
					pos += mlen;
				} else { literal(pos); ++pos; }
			}
			ssg_wave_ldssync();
			for (int q = s0; q < s1 && q + 4 <= n; ++q) {   /* the region enters the table: every position, the two newest per hash */
				const uint32_t hq = ((bz_load32(src + q) * 2654435761u) >> (32 - BZ_HBITS)) & ~1u;
				ht[hq + 1] = ht[hq]; ht[hq] = (uint16_t)q;
			}
			ssg_wave_ldssync();
		}
		chunk_syms[j] = (uint16_t)(ns - ns0);
	}
	ssg_wave_ldssync();
	/* ---- 2. the three codes, by one lane ---- */
	if (lane == 0) {
		f_ll[256] = 1;   /* end of block */
		bz_code_lengths(f_ll, 286, 15, l_ll, w_idx, w_wt, w_par);
		bz_code_lengths(f_d, 30, 15, l_d, w_idx, w_wt, w_par);
		int hlit = 286; while (hlit > 257 && l_ll[hlit - 1] == 0) --hlit;
		int hdist = 30; while (hdist > 1 && l_d[hdist - 1] == 0) --hdist;
		/* run-length form of the two length sequences (RFC 1951 3.2.7: 16 = repeat previous 3-6, 17 = 3-10 zeros, 18 = 11-138 zeros) */
		int nr = 0;
		for (int t = 0; t < 2; ++t) {
			const uint8_t *L = t ? l_d : l_ll; const int cnt = t ? hdist : hlit;
			for (int i = 0; i < cnt; ) {
				const int v = L[i]; int run = 1;
				while (i + run < cnt && L[i + run] == v) ++run;
				i += run;
				if (v == 0) {
					while (run >= 11) { const int r = run < 138 ? run : 138; rle[nr++] = (uint16_t)(18 | (r - 11) << 5); run -= r; }
					if (run >= 3) { rle[nr++] = (uint16_t)(17 | (run - 3) << 5); run = 0; }
					while (run-- > 0) rle[nr++] = 0;
				} else {
					rle[nr++] = (uint16_t)v; --run;
					while (run >= 3) { const int r = run < 6 ? run : 6; rle[nr++] = (uint16_t)(16 | (r - 3) << 5); run -= r; }
					while (run-- > 0) rle[nr++] = (uint16_t)v;
				}
			}
		}
		for (int k = 0; k < nr; ++k) ++f_cl[rle[k] & 31];
		bz_code_lengths(f_cl, 19, 7, l_cl, w_idx, w_wt, w_par);
		bz_codes(l_ll, 286, 15, c_ll); bz_codes(l_d, 30, 15, c_d); bz_codes(l_cl, 19, 7, c_cl);
		const int order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		int hclen = 19; while (hclen > 4 && l_cl[order[hclen - 1]] == 0) --hclen;
		uint32_t hb = 3 + 5 + 5 + 4 + 3 * (uint32_t)hclen;
		for (int k = 0; k < nr; ++k) { const int s = rle[k] & 31; hb += l_cl[s] + (s == 16 ? 2 : s == 17 ? 3 : s == 18 ? 7 : 0); }
		sh_misc[0] = (uint32_t)nr; sh_misc[1] = (uint32_t)hlit; sh_misc[2] = (uint32_t)hdist; sh_misc[3] = (uint32_t)hclen; sh_misc[4] = hb;
	}
	ssg_wave_ldssync();
	/* ---- 3. sizes, placement, bits: the stream is the header, then the chunks in the order of their positions (region by region, lane by lane) ---- */
	auto sym_bits = [&](const uint32_t s) -> uint32_t {
		if (!(s & 0x80000000u)) return l_ll[s];
		int lc, le, lv, dc, de, dv;
		bz_len_code((int)((s >> 16) & 0x1ff), lc, le, lv); bz_dist_code((int)(s & 0xffff) + 1, dc, de, dv);
		return (uint32_t)(l_ll[257 + lc] + le + l_d[dc] + de);
	};
	const uint32_t hb = sh_misc[4];
	uint32_t chunk_bit0[BZ_MAX_REGIONS], base = hb;
	{	int k = 0;
		SSG_UNROLL for (int j = 0; j < BZ_MAX_REGIONS; ++j) {
			uint32_t bits = 0;
			for (int e = k + chunk_syms[j]; k < e; ++k) bits += sym_bits(my_sym[(size_t)k * 64]);
			const uint32_t incl = (uint32_t)wv_scan_add((int)bits);
			chunk_bit0[j] = base + incl - bits;
			base += (uint32_t)wv_get((int)incl, 63);
		}
	}
	const uint32_t total_bits = base + l_ll[256];
	const uint32_t total_bytes = (total_bits + 7) >> 3;
	if (total_bytes >= (uint32_t)n + 5u) {   /* does not shrink: one stored block (RFC 1951 3.2.4) */
		uint8_t *o = (uint8_t*)out;
		if (lane == 0) { o[0] = 1; o[1] = (uint8_t)(n & 0xff); o[2] = (uint8_t)(n >> 8); o[3] = (uint8_t)~o[1]; o[4] = (uint8_t)~o[2]; }
		for (int k = lane; k < n; k += 64) o[5 + k] = src[k];
		if (lane == 0) size[b] = (uint32_t)n + 5u;
		return;
	}
	bz_writer_t w;
	if (lane == 0) {   /* block header: BFINAL = 1, BTYPE = 2, HLIT, HDIST, HCLEN, the code-length code, the two length sequences; and the end-of-block code */
		const int order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		bz_w_init(w, out, 0);
		bz_w_put(w, 1u | 2u << 1, 3);
		bz_w_put(w, sh_misc[1] - 257, 5); bz_w_put(w, sh_misc[2] - 1, 5); bz_w_put(w, sh_misc[3] - 4, 4);
		for (uint32_t k = 0; k < sh_misc[3]; ++k) bz_w_put(w, l_cl[order[k]], 3);
		for (uint32_t k = 0; k < sh_misc[0]; ++k) {
			const int s = rle[k] & 31, ev = rle[k] >> 5;
			bz_w_put(w, c_cl[s], l_cl[s]);
			if (s >= 16) bz_w_put(w, (uint32_t)ev, s == 16 ? 2 : s == 17 ? 3 : 7);
		}
		bz_w_end(w);
		bz_w_init(w, out, base); bz_w_put(w, c_ll[256], l_ll[256]); bz_w_end(w);
	}
	{	int k = 0;
		SSG_UNROLL for (int j = 0; j < BZ_MAX_REGIONS; ++j) if (chunk_syms[j]) {
			bz_w_init(w, out, chunk_bit0[j]);
			for (int e = k + chunk_syms[j]; k < e; ++k) {
				const uint32_t s = my_sym[(size_t)k * 64];
				if (s & 0x80000000u) {
					int lc, le, lv, dc, de, dv;
					bz_len_code((int)((s >> 16) & 0x1ff), lc, le, lv); bz_dist_code((int)(s & 0xffff) + 1, dc, de, dv);
					bz_w_put(w, c_ll[257 + lc], l_ll[257 + lc]); if (le) bz_w_put(w, (uint32_t)lv, le);
					bz_w_put(w, c_d[dc], l_d[dc]); if (de) bz_w_put(w, (uint32_t)dv, de);
				} else bz_w_put(w, c_ll[s], l_ll[s]);
			}
			bz_w_end(w);
		}
	}
	if (lane == 0) size[b] = total_bytes;
}

/* the blocks' streams from their strided temporary places to one dense buffer: off[b] .. off[b+1] */
__global__ void __launch_bounds__(256) ssg_k_bgzf_compact(const uint8_t *tmp, const uint64_t *off, int n_blocks, uint8_t *dense)
{
	const int b = (int)blockIdx.x;
	if (b >= n_blocks) return;
	const uint8_t *s = tmp + (size_t)b * BZ_OUT_STRIDE; uint8_t *d = dense + off[b];
	const uint32_t n = (uint32_t)(off[b + 1] - off[b]);
	for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) d[k] = s[k];
}
#endif
