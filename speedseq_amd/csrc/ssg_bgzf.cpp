/*
 * ssg_bgzf.cpp -- BGZF deflate on the device (k_bgzf.h): the host entry point of `sambamba sort`'s last
 * step.  A translation unit of its own (seconds to compile, variants by `make variant VUNITS=ssg_bgzf`).
 */
#include <algorithm>
#include <vector>
#include "ssg_rt.h"
#include "k_bgzf.h"
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"

SSG_ABI_FP_DEFINE(bgzf)
#define CHK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

extern "C" {

/* page-locked host memory from the library's pool (ssg_rt.h): copies to and from the device run at bus speed out of these */
void *ssg_host_alloc(size_t n) { return rt_host_alloc(n); }
void ssg_host_free(void *p) { rt_host_free(p); }

int ssg_bgzf_deflate(const uint8_t *payload, const uint64_t *cut, long n_blocks, uint8_t *out, uint64_t out_cap, uint64_t *out_off)
{
	if (rt_device_count() < 1) { ssg_err_msg = "no HIP device visible: libssgpu has no CPU path"; return SSG_ENODEV; }
	out_off[0] = 0;
	if (n_blocks <= 0) return 0;
	for (long b = 0; b < n_blocks; ++b) if (cut[b + 1] < cut[b] || cut[b + 1] - cut[b] > BZ_MAX_PAYLOAD) { ssg_err_msg = "ssg_bgzf_deflate: a block's payload exceeds 0xff00 bytes"; return SSG_EINVAL; }
	const long BB = 4096;   /* blocks per device call: 256 MB of temporary output, 1 GB of symbol lists */
	const long nbmax = std::min(BB, n_blocks);
	dbuf<uint8_t> d_pay((size_t)nbmax * BZ_MAX_PAYLOAD + 8), d_tmp((size_t)nbmax * BZ_OUT_STRIDE), d_dense((size_t)nbmax * BZ_OUT_STRIDE);
	dbuf<uint32_t> d_sym((size_t)nbmax * BZ_STRETCH_CAP * 64), d_size(nbmax);
	dbuf<uint64_t> d_cut(nbmax + 1), d_off(nbmax + 1);
	if (!d_pay.ok() || !d_tmp.ok() || !d_dense.ok() || !d_sym.ok() || !d_size.ok() || !d_cut.ok() || !d_off.ok()) { ssg_err_msg = "device allocation failed: BGZF deflate"; return SSG_ENOMEM; }
	std::vector<uint64_t> rel((size_t)nbmax + 1), off((size_t)nbmax + 1); std::vector<uint32_t> sz((size_t)nbmax);
	for (long b0 = 0; b0 < n_blocks; b0 += BB) {
		const long nb = std::min(BB, n_blocks - b0);
		const uint64_t base = cut[b0], bytes = cut[b0 + nb] - base;
		for (long k = 0; k <= nb; ++k) rel[(size_t)k] = cut[b0 + k] - base;
		CHK(rt_h2d(d_pay.p, payload + base, bytes)); CHK(d_cut.up(rel.data(), (size_t)nb + 1));
		SSG_LAUNCH(ssg_k_bgzf_deflate, nb, 64, 0, (const uint8_t*)d_pay.p, (const uint64_t*)d_cut.p, (int)nb, d_tmp.p, d_sym.p, d_size.p);
		CHK(rt_sync());
		CHK(d_size.down(sz.data(), (size_t)nb));
		off[0] = 0; for (long k = 0; k < nb; ++k) off[(size_t)k + 1] = off[(size_t)k] + sz[(size_t)k];
		if (out_off[b0] + off[(size_t)nb] > out_cap) { ssg_err_msg = "ssg_bgzf_deflate: output buffer too small"; return SSG_EOVERFLOW; }
		CHK(d_off.up(off.data(), (size_t)nb + 1));
		SSG_LAUNCH(ssg_k_bgzf_compact, nb, 256, 0, (const uint8_t*)d_tmp.p, (const uint64_t*)d_off.p, (int)nb, d_dense.p);
		CHK(rt_sync());
		CHK(rt_d2h(out + out_off[b0], d_dense.p, off[(size_t)nb]));
		for (long k = 0; k < nb; ++k) out_off[b0 + k + 1] = out_off[b0] + off[(size_t)k + 1];
	}
	return 0;
}

} /* extern "C" */
