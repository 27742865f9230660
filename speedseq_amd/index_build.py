"""FM-index construction on the GPU (`bwa index`), PyTorch used as plumbing only
(device memory + radix sort): prefix-doubling suffix array of T = fwd || revcomp(fwd), BWT with the
interleaved Occ checkpoints in upstream's on-disk layout (one 64-byte block = 4 x u64 counts +
8 x u32 symbols), sampled SA (interval 32) and the 2-bit .pac.

The arrays are bit-identical to what upstream `bwa index` writes (checked against the bundled
example index in tests/test_index_build.py); bench.py uses this to index its synthetic reference in
HBM without ever touching the host.  Limit: 2*l_pac < 2^31 (about 1.07 Gbp) in this round.
"""
import numpy as np
import torch


def _ranks_from_sorted(keys_sorted):
    d = torch.ones_like(keys_sorted, dtype=torch.int64)
    d[0] = 0
    d[1:] = (keys_sorted[1:] != keys_sorted[:-1]).to(torch.int64)
    return torch.cumsum(d, 0)


def suffix_array(T):
    """Suffix array of T (uint8 codes 0..3) with an implicit smallest terminator; int64 tensor."""
    n = T.numel()
    dev = T.device
    assert n < (1 << 31)
    K = 21
    pad = torch.zeros(n + K, dtype=torch.int64, device=dev)
    pad[:n] = T.to(torch.int64) + 1
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for k in range(K):
        key = key * 8 + pad[k:k + n]
    del pad
    ks, sa = torch.sort(key)
    del key
    rs = _ranks_from_sorted(ks)
    del ks
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    rank[sa] = rs
    h = K
    while int(rs[-1]) != n - 1:
        nxt = torch.zeros(n, dtype=torch.int64, device=dev)
        if h < n:
            nxt[:n - h] = rank[h:] + 1
        key = (rank << 32) | nxt
        del nxt
        ks, sa = torch.sort(key)
        del key
        rs = _ranks_from_sorted(ks)
        del ks
        rank[sa] = rs
        h *= 2
    return sa


def build_index_arrays(fwd_codes):
    """fwd_codes: uint8 tensor (0..3) of the concatenated forward reference (no N).
    Returns dict of tensors/ints in upstream's layouts: bwt (u32 words incl. Occ), primary, L2[5],
    sa (sampled, int64, sa[0] = -1), pac (uint8)."""
    dev = fwd_codes.device
    l_pac = fwd_codes.numel()
    T = torch.cat([fwd_codes, (3 - fwd_codes).flip(0)])
    n = T.numel()
    sa = suffix_array(T)
    # BWT: row 0 ("$") is preceded by T[n-1]; the row of suffix 0 (primary) is dropped
    primary = int(torch.nonzero(sa == 0)[0, 0]) + 1
    prev = torch.where(sa > 0, sa - 1, torch.zeros_like(sa))
    Bfull = T[prev]                                  # rows 1..n
    keep = sa != 0
    B = torch.cat([T[n - 1:n], Bfull[keep]])         # length n
    del Bfull, keep, prev
    cnt = torch.bincount(T.to(torch.int64), minlength=4)
    L2 = [0] + [int(x) for x in torch.cumsum(cnt, 0)]
    # sampled SA: with-$ row r (1..n) = sa[r-1]; keep rows that are multiples of 32
    n_sa = (n + 32) // 32
    samp = torch.full((n_sa,), -1, dtype=torch.int64, device=dev)
    samp[1:] = sa[torch.arange(1, n_sa, device=dev) * 32 - 1]
    del sa
    # Occ checkpoints + 2-bit packing, 128 symbols per block
    nblk = (n + 127) // 128
    Bp = torch.zeros(nblk * 128, dtype=torch.uint8, device=dev)
    Bp[:n] = B
    valid = torch.zeros(nblk * 128, dtype=torch.bool, device=dev)
    valid[:n] = True
    blk = Bp.view(nblk, 128)
    vblk = valid.view(nblk, 128)
    counts = torch.zeros((nblk + 1, 4), dtype=torch.int64, device=dev)
    for c in range(4):
        per = ((blk == c) & vblk).sum(1)
        counts[1:, c] = torch.cumsum(per, 0)
    w = blk.view(nblk, 8, 16).to(torch.int64)
    shifts = torch.arange(15, -1, -1, device=dev, dtype=torch.int64) * 2
    words = (w << shifts).sum(2)                     # (nblk, 8) values < 2^32
    # file layout: per block 8 count words (4 x u64 LE) + symbol words; final counts after the last symbol word
    n_words = (n + 15) // 16
    total = n_words + 8 * (nblk + 1)
    out = torch.zeros(nblk * 16 + 16, dtype=torch.int64, device=dev)
    o2 = out[:nblk * 16].view(nblk, 16)
    o2[:, 0:8:2] = counts[:nblk] & 0xffffffff
    o2[:, 1:8:2] = counts[:nblk] >> 32
    o2[:, 8:16] = words
    last_sym_words = n_words - (nblk - 1) * 8        # symbol words in the last block (1..8)
    tail = (nblk - 1) * 16 + 8 + last_sym_words
    out[tail:tail + 8:2] = counts[nblk] & 0xffffffff
    out[tail + 1:tail + 8:2] = counts[nblk] >> 32
    bwt = out[:total] & 0xffffffff
    # .pac: 4 bases per byte, MSB first
    npb = l_pac // 4 + 1
    fp = torch.zeros(npb * 4, dtype=torch.int64, device=dev)
    fp[:l_pac] = fwd_codes.to(torch.int64)
    pac = (fp.view(npb, 4) << torch.tensor([6, 4, 2, 0], device=dev, dtype=torch.int64)).sum(1).to(torch.uint8)
    return {"bwt": bwt, "primary": primary, "L2": L2, "sa": samp, "pac": pac, "l_pac": l_pac, "n": n}


def arrays_to_numpy(ix):
    """Host copies in the dtypes ssg_index_from_arrays expects."""
    bwt = ix["bwt"].cpu().numpy().astype(np.uint32)
    sa = ix["sa"].cpu().numpy().astype(np.int64).view(np.uint64)
    pac = ix["pac"].cpu().numpy()
    return bwt, sa, pac


def write_index_files(prefix, ix, names, lens):
    """Write prefix.{bwt,sa,pac,ann,amb} exactly as upstream `bwa index` would."""
    bwt, sa, pac = arrays_to_numpy(ix)
    l_pac = ix["l_pac"]
    with open(prefix + ".bwt", "wb") as f:
        f.write(np.array([ix["primary"]] + ix["L2"][1:], dtype=np.uint64).tobytes())
        f.write(bwt.tobytes())
    with open(prefix + ".sa", "wb") as f:
        f.write(np.array([ix["primary"]] + ix["L2"][1:] + [32, ix["n"]], dtype=np.uint64).tobytes())
        f.write(sa[1:].tobytes())
    with open(prefix + ".pac", "wb") as f:
        nb = l_pac // 4 + (1 if l_pac % 4 else 0)
        f.write(pac[:nb].tobytes())
        if l_pac % 4 == 0:
            f.write(b"\0")
        f.write(bytes([l_pac % 4]))
    with open(prefix + ".ann", "w") as f:
        f.write("%d %d %d\n" % (l_pac, len(names), 11))
        o = 0
        for nm, ln in zip(names, lens):
            f.write("0 %s (null)\n%d %d 0\n" % (nm, o, ln))
            o += ln
    with open(prefix + ".amb", "w") as f:
        f.write("%d %d 0\n" % (l_pac, len(names)))
