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


def global_markdup(sig, valid, ordinal, device="cpu"):
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
    # owner side: minimum ordinal per signature
    keys, inv = torch.unique(recv[:, :3], dim=0, return_inverse=True)
    first = torch.full((keys.shape[0],), torch.iinfo(torch.int64).max, dtype=torch.int64, device=device)
    first = first.scatter_reduce(0, inv, recv[:, 3], reduce="amin")
    verdict = ((recv[:, 3] > first[inv]) & (recv[:, 4] != 0)).to(torch.uint8)
    back = torch.empty(n, dtype=torch.uint8, device=device)
    dist.all_to_all_single(back, verdict.contiguous(), output_split_sizes=sc, input_split_sizes=rc)
    dup = torch.empty(n, dtype=torch.uint8, device=device)
    dup[order] = back
    return dup


def max_over_ranks(seconds, device="cpu"):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
