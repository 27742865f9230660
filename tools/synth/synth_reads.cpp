// Soak utility (not part of the product library): interleaved FASTQ records of simulated read pairs in ONE kernel launch per chunk, so that the
// generator of a whole-genome soak (tools/soak.py --stream: 400 M pairs = 125 GB of FASTQ through a FIFO) costs the device milliseconds per
// million pairs and does not compete with the pipeline it feeds (the torch simulator of bench.py took 0.5 s of device time per million).
// The model is bench.py simulate_pairs' (SURVEY.md 8d): fragments ~ N(mean, std) placed on contigs in proportion to their lengths; 5 % exact fragment
// duplicates (a pair re-draws the fragment of an EARLIER pair of the whole stream, fresh errors), 1 % long inserts (5-50 kb), 1 % chimeric first
// reads, either strand; per read 7 % with one indel of 1-5 bases, 0.5 % substitutions, 0.1 % N.  Every pair is a pure function of (seed, its index in
// the stream): chunks can be produced in any order and a duplicate's source can lie in another chunk.
// Record layout = bench.py write_fastq: "@p%09d\n" SEQ "\n+\n" 'I' x rl "\n", two records per pair.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/synth/synth_reads.cpp -o tools/synth/libsynthreads.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t sr64(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
struct rng_t { uint64_t s; __device__ uint64_t next() { s = sr64(s); return s; } __device__ double u() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); } };

struct frag_t { int64_t g; int32_t d; };   // global start of the fragment (forward strand), its length

// the fragment pair `gp` would have if it were no duplicate: contig ~ length, insert ~ N(mean, std) (Box-Muller), or a long one
__device__ frag_t base_fragment(uint64_t seed, int64_t gp, const int64_t *ctg_off, const int64_t *ctg_len, int n_ctg, int64_t total, int rl, int ins_mean, int ins_std, double disc_lo, double disc_hi)
{
	rng_t r = { sr64(seed ^ sr64((uint64_t)gp * 2 + 1)) };
	const double u = r.u();
	int64_t x = (int64_t)(r.u() * (double)total);
	int lo = 0, hi = n_ctg - 1;
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ctg_off[mid] <= x) lo = mid; else hi = mid - 1; }
	const int64_t cl = ctg_len[lo];
	const double u1 = r.u(), u2 = r.u();
	int64_t d = (int64_t)(sqrt(-2.0 * log(u1 > 1e-300 ? u1 : 1e-300)) * cos(6.283185307179586 * u2) * ins_std + ins_mean);
	if (u >= disc_lo && u < disc_hi) d = 5000 + (int64_t)(r.u() * 45000.0);
	const int need = rl + 8;
	if (d > cl - 1) d = cl - 1;
	if (d < need + 1) d = need + 1;
	int64_t span = cl - d; if (span < 1) span = 1;
	frag_t f; f.g = ctg_off[lo] + (int64_t)(r.u() * (double)span); f.d = (int32_t)d;
	if (f.g + f.d > ctg_off[lo] + cl) f.g = ctg_off[lo] + (cl > f.d ? cl - f.d : 0);
	return f;
}

__global__ void __launch_bounds__(256) k_fastq_pairs(const uint8_t *ref, const int64_t *ctg_off, const int64_t *ctg_len, int n_ctg, int64_t total, int64_t first_pair, int n_pairs, int rl,
                              uint64_t seed, int ins_mean, int ins_std, uint8_t *out)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= 2L * n_pairs) return;
	const int64_t gp = first_pair + (t >> 1); const int which = (int)(t & 1);     // one lane per read
	const double dup_frac = 0.05, disc_frac = 0.01, chim_frac = 0.01;
	rng_t rp = { sr64(seed ^ sr64((uint64_t)gp * 2)) };                             // the pair's own stream (both lanes of the pair draw the same)
	const double u = rp.u();
	int64_t src = gp;
	if (u < dup_frac && gp > 0) src = (int64_t)(rp.u() * (double)gp);                // the fragment of an earlier pair, errors of its own
	const frag_t f = base_fragment(seed, src, ctg_off, ctg_len, n_ctg, total, rl, ins_mean, ins_std, dup_frac, dup_frac + disc_frac);
	const bool chim = u >= dup_frac + disc_frac && rp.u() < chim_frac;
	const int64_t other = (int64_t)(rp.u() * (double)(total - rl - 16));
	const int bp = 40 + (int)(rp.u() * (double)(rl - 80 > 1 ? rl - 80 : 1));
	const bool flip = rp.u() < 0.5;
	// read `which` of the pair: the forward end or the reverse-complemented far end (swapped when the pair is flipped)
	const bool far_end = (which == 1) != flip;
	rng_t rr = { sr64(seed ^ sr64(((uint64_t)gp * 2 + (uint64_t)which) ^ 0x5bd1e995u)) };
	const bool has_indel = rr.u() < 0.07;
	const int ip = 5 + (int)(rr.u() * (double)(rl - 10 > 1 ? rl - 10 : 1)), il = 1 + (int)(rr.u() * 5.0);
	const bool is_del = rr.u() < 0.5;
	const int rec = 1 + 10 + 1 + rl + 3 + rl + 1;
	uint8_t *o = out + t * (int64_t)rec;
	o[0] = '@'; o[1] = 'p';
	{ int64_t v = gp; for (int k = 8; k >= 0; --k) { o[2 + k] = (uint8_t)('0' + v % 10); v /= 10; } }
	o[11] = '\n';
	uint8_t *s = o + 12;
	for (int j = 0; j < rl; ++j) {
		int sj = j;                                                                   // column of the error-free read this base comes from
		bool inserted = false;
		if (has_indel) { if (is_del) { if (j >= ip) sj = j + il; } else { if (j >= ip + il) sj = j - il; else if (j >= ip) inserted = true; } }
		int b;
		if (inserted) b = (int)(rr.next() >> 62);
		else {
			if (!far_end) b = ref[((chim && sj >= bp) ? other : f.g) + sj];              // the forward end; a chimeric one continues at another locus
			else b = 3 - ref[f.g + f.d - 1 - sj];
		}
		const uint64_t e = rr.next();
		if ((e & 0xffff) < 328) b = (b + 1 + (int)((e >> 16) % 3)) & 3;                  // 0.5 % substitutions
		if (((e >> 32) & 0xffff) < 66) b = 4;                                          // 0.1 % N
		s[j] = (uint8_t)"ACGTN"[b];
	}
	s[rl] = '\n'; s[rl + 1] = '+'; s[rl + 2] = '\n';
	for (int j = 0; j < rl; ++j) s[rl + 3 + j] = 'I';
	s[2 * rl + 3] = '\n';
}

// d_ref: forward-strand codes 0..3 of all contigs back to back (total bases); d_ctg_off / d_ctg_len: device arrays of n_ctg entries;
// d_out: 2 * n_pairs * (2 * rl + 16) bytes.  Returns 0 / -1.
extern "C" int synth_fastq_pairs(const uint8_t *d_ref, const int64_t *d_ctg_off, const int64_t *d_ctg_len, int n_ctg, int64_t total, int64_t first_pair, int n_pairs, int rl,
                                 uint64_t seed, int ins_mean, int ins_std, uint8_t *d_out)
{
	if (n_pairs <= 0) return 0;
	hipLaunchKernelGGL(k_fastq_pairs, dim3((unsigned)((2L * n_pairs + 255) / 256)), dim3(256), 0, 0, d_ref, d_ctg_off, d_ctg_len, n_ctg, total, first_pair, n_pairs, rl, seed, ins_mean, ins_std, d_out);
	return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : -1;
}
