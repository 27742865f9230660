"""Multi-GPU layer of the hot path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Alignment needs no collective: whole upstream batches (the scope of the insert-size model) are dealt
round-robin to ranks, each with a full index replica.  Duplicate marking is the one real exchange
step: samblaster keeps the FIRST pair of every signature in input order, so signatures travel to an
owner rank chosen by hash (all-to-all, 32 bytes per pair, every xGMI link busy at once -- no ring),
the owner keeps the minimum global ordinal per signature, and 1-byte verdicts travel back.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_batches(n_batches, rank, world):
    """Upstream batches owned by `rank` (round-robin): batch composition is preserved, so the
    per-batch insert-size statistics are identical to a single-process run."""
    return list(range(rank, n_batches, world))


def signatures(ends):
    """5'-unclipped pair signatures (int64 [n,3]) + validity mask from SBL_END_DT records, same
    arithmetic as ssg_k_sig (csrc/k_misc.h)."""
    seq = torch.as_tensor(ends["seq"].astype(np.int64)).view(-1, 2)
    pos = torch.as_tensor(ends["pos"].astype(np.int64)).view(-1, 2)
    flag = torch.as_tensor(ends["flag"].astype(np.int64)).view(-1, 2)
    lclip = torch.as_tensor(ends["lclip"].astype(np.int64)).view(-1, 2)
    rclip = torch.as_tensor(ends["rclip"].astype(np.int64)).view(-1, 2)
    ralen = torch.as_tensor(ends["ralen"].astype(np.int64)).view(-1, 2)
    mapped = ((flag & 4) == 0) & (seq >= 0)
    strand = ((flag & 0x10) != 0).to(torch.int64)
    p5 = torch.where(strand == 1, pos + ralen - 1 + rclip, pos - lclip) + (1 << 31)
    k = torch.stack([seq, p5, strand], 2)                          # [n, 2, 3]
    a, b = k[:, 0], k[:, 1]
    swap = (a[:, 0] > b[:, 0]) | ((a[:, 0] == b[:, 0]) & ((a[:, 1] > b[:, 1]) | ((a[:, 1] == b[:, 1]) & (a[:, 2] > b[:, 2]))))
    lo = torch.where(swap[:, None], b, a)
    hi = torch.where(swap[:, None], a, b)
    both = mapped[:, 0] & mapped[:, 1]
    one = mapped[:, 0] ^ mapped[:, 1]
    single = torch.where(mapped[:, 0:1], a, b)
    sig = torch.zeros(len(seq), 3, dtype=torch.int64)
    sig[both] = torch.stack([(lo[:, 0] << 32) | hi[:, 0], (lo[:, 1] << 1) | lo[:, 2], (hi[:, 1] << 1) | hi[:, 2]], 1)[both]
    sig[one] = torch.stack([single[:, 0], (single[:, 1] << 1) | single[:, 2], torch.zeros_like(single[:, 0])], 1)[one]
    return sig, both | one


def _mix(x):
    x = x ^ (x >> 33)
    x = x * -49064778989728563            # 0xff51afd7ed558ccd as int64
    x = x ^ (x >> 33)
    return x


OWNER_PATH = ["torch"]        # which implementation took the owner-side decisions last (bench.py reports it)


def _owner_verdicts(recv, lib):
    """owner side: an element is a duplicate iff the same signature arrived with a smaller global ordinal.  With libssgpu (device tensors)
    this is ssg_markdup_sig_dev -- the kernels of the single-GPU duplicate marking (radix sort by ordinal, hash sort, run scan); without
    it (the gloo tests on CPU tensors) the same rule in torch."""
    if lib is not None and recv.is_cuda:
        from . import capi
        n = recv.shape[0]
        sig = recv[:, :3].clone()
        sig[recv[:, 4] == 0] = -1                      # not valid: never a duplicate (all ones)
        sig = sig.contiguous(); ordi = recv[:, 3].contiguous()
        verdict = torch.empty(n, dtype=torch.uint8, device=recv.device)
        try:
            if n:
                torch.cuda.current_stream().synchronize()   # libssgpu launches on the default stream
                capi.markdup_sig_dev(lib, n, sig.data_ptr(), ordi.data_ptr(), verdict.data_ptr())
            OWNER_PATH[0] = "libssgpu (ssg_markdup_sig_dev)"
            return verdict
        except Exception as e:   # the same rule in torch below; said out loud, never silently
            import sys
            sys.stderr.write("[dist] ssg_markdup_sig_dev failed (%r): owner-side duplicate marking falls back to torch\n" % (e,))
            OWNER_PATH[0] = "torch (libssgpu path failed: %r)" % (e,)
    keys, inv = torch.unique(recv[:, :3], dim=0, return_inverse=True)
    first = torch.full((keys.shape[0],), torch.iinfo(torch.int64).max, dtype=torch.int64, device=recv.device)
    first = first.scatter_reduce(0, inv, recv[:, 3], reduce="amin")
    return ((recv[:, 3] > first[inv]) & (recv[:, 4] != 0)).to(torch.uint8)


def global_markdup(sig, valid, ordinal, device="cpu", lib=None):
    """Exact first-seen-wins duplicate flags across all ranks.  sig int64 [n,3], valid bool [n],
    ordinal int64 [n] = global input index of each local pair.  Returns uint8 [n]."""
    world, n = dist.get_world_size(), sig.shape[0]
    sig, valid, ordinal = sig.to(device), valid.to(device), ordinal.to(device)
    h = _mix(sig[:, 0] ^ _mix(sig[:, 1] ^ _mix(sig[:, 2])))
    dest = torch.where(valid, (h & 0x7fffffffffffffff) % world, torch.zeros_like(h))
    order = torch.argsort(dest, stable=True)
    payload = torch.cat([sig, ordinal[:, None], valid[:, None].to(torch.int64)], 1)[order].contiguous()
    send_counts = torch.bincount(dest, minlength=world)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    recv = torch.empty(sum(rc), 5, dtype=torch.int64, device=device)
    dist.all_to_all_single(recv, payload, output_split_sizes=rc, input_split_sizes=sc)
    verdict = _owner_verdicts(recv, lib)
    back = torch.empty(n, dtype=torch.uint8, device=device)
    dist.all_to_all_single(back, verdict.contiguous(), output_split_sizes=sc, input_split_sizes=rc)
    dup = torch.empty(n, dtype=torch.uint8, device=device)
    dup[order] = back
    return dup


def max_over_ranks(seconds, device="cpu"):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def coordinate_range_exchange(keys, ordinal, recs, device="cpu", n_samples=1024):
    """Coupling 3 of SURVEY.md 8e, the one exchange the north star names: the coordinate-sorted merge as a SAMPLE SORT, not a
    gather.  Every rank holds records with their samtools sort key (bam_sort.c:1607-1614: tid<<32 | (pos+1)<<1 | reverse) and
    global input ordinal.  Ranks all-gather a sample of their keys, agree on world-1 splitters, send every record to the rank
    that owns its key range (one all-to-all over all xGMI links at once), and sort what they received by (key, ordinal) --
    rank r then holds the r-th contiguous stretch of the coordinate-sorted file, ties in input order exactly as samtools'
    stable sort / merge leaves them.
    keys: int64 [n] (the 64-bit key reinterpreted), ordinal: int64 [n], recs: uint8 [n, R].  Returns (keys, ordinal, recs) of this
    rank's stretch, sorted, and the number of payload bytes this rank sent to other ranks."""
    world, rank = dist.get_world_size(), dist.get_rank()
    keys, ordinal, recs = keys.to(device), ordinal.to(device), recs.to(device)
    n = keys.numel()
    # unsigned order on the int64 view: flip the sign bit
    ukey = keys ^ torch.iinfo(torch.int64).min
    if world > 1:
        srt = torch.sort(ukey).values
        pick = (torch.arange(n_samples, device=device, dtype=torch.int64) * max(n, 1)) // n_samples
        sample = srt[pick.clamp(max=max(n - 1, 0))] if n else torch.full((n_samples,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=device)
        allsamp = [torch.empty_like(sample) for _ in range(world)]
        dist.all_gather(allsamp, sample)
        allsamp = torch.sort(torch.cat(allsamp)).values
        splitters = allsamp[(torch.arange(1, world, device=device) * allsamp.numel()) // world]
        dest = torch.bucketize(ukey, splitters, right=False)      # equal keys share a destination: ties never straddle two ranks
    else:
        dest = torch.zeros(n, dtype=torch.int64, device=device)
    order = torch.argsort(dest, stable=True)
    send_counts = torch.bincount(dest, minlength=world)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    meta = torch.stack([ukey, ordinal], 1)[order].contiguous()
    meta_r = torch.empty(sum(rc), 2, dtype=torch.int64, device=device)
    dist.all_to_all_single(meta_r, meta, output_split_sizes=rc, input_split_sizes=sc)
    recs_s = recs[order].contiguous()
    recs_r = torch.empty((sum(rc),) + tuple(recs.shape[1:]), dtype=recs.dtype, device=device)
    dist.all_to_all_single(recs_r, recs_s, output_split_sizes=rc, input_split_sizes=sc)
    # local order: by key, ties by global input ordinal (two stable passes)
    o1 = torch.argsort(meta_r[:, 1], stable=True)
    o2 = torch.argsort(meta_r[o1, 0], stable=True)
    perm = o1[o2]
    sent = (n - sc[rank]) * (16 + int(np.prod(recs.shape[1:])) * recs.element_size())
    return meta_r[perm, 0] ^ torch.iinfo(torch.int64).min, meta_r[perm, 1], recs_r[perm], sent
