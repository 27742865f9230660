"""N>1 path on CPU: world_size-2 `gloo` run of the multi-GPU layer (speedseq_amd/dist.py).
Pairs are dealt to two ranks by upstream batch; the exact global duplicate flags computed through the
signature all-to-all must equal the oracle samblaster's single-stream flags."""
import os
import subprocess
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common
from common import ROOT


def _worker(rank, world, port, ends_path, batch_path, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speedseq_amd import capi, dist as sdist
    ends = np.load(ends_path)
    pair_batch = np.load(batch_path)
    n_batches = int(pair_batch.max()) + 1
    mine = np.isin(pair_batch, sdist.shard_batches(n_batches, rank, world))
    ordinal = torch.as_tensor(np.nonzero(mine)[0].astype(np.int64))
    my_ends = ends.reshape(-1, 2)[mine].reshape(-1)
    sig, valid = sdist.signatures(my_ends)
    dup = sdist.global_markdup(sig, valid, ordinal)
    t = sdist.max_over_ranks(1.0 + rank)
    assert t == float(world)
    np.save(os.path.join(out_dir, "dup%d.npy" % rank), np.stack([ordinal.numpy(), dup.numpy().astype(np.int64)]))
    dist.barrier()
    dist.destroy_process_group()


def test_global_dedup_two_ranks_gloo(oracle):
    oidx = oracle.idx_load(common.EXAMPLE_FA)
    n_pairs = 600
    pairs, seqs, seq, off = common.sim_reads(n_pairs, 31, dup_frac=0.25)
    names = []
    for nm, _, _ in pairs:
        names += [nm, nm]
    otext, _, _ = oracle.process_pairs(oidx, seq, off, names, None, 0, "", 4)
    oflags, _ = common.oracle_dup_flags(oracle, otext, "@SQ\tSN:20_slice\tLN:321635\n")
    ends = common.sam_primary_ends(otext, ["20_slice"])
    pair_batch = (np.arange(n_pairs) // 50).astype(np.int32)       # 12 "upstream batches"
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "ends.npy"), ends)
        np.save(os.path.join(d, "pb.npy"), pair_batch)
        port = 29500 + os.getpid() % 2000
        mp.spawn(_worker, args=(2, port, os.path.join(d, "ends.npy"), os.path.join(d, "pb.npy"), d), nprocs=2, join=True)
        got = np.zeros(n_pairs, dtype=np.uint8)
        seen = np.zeros(n_pairs, dtype=bool)
        for r in range(2):
            o, v = np.load(os.path.join(d, "dup%d.npy" % r))
            got[o] = v
            seen[o] = True
    assert seen.all()
    assert np.array_equal(got, oflags), (int(got.sum()), int(oflags.sum()))
    assert got.sum() > 50


def test_device_signatures_through_the_exchange(emu_lib, oracle):
    """The N>1 bench path: ssg_hotpath_dev_sig's signatures (device layout) through dist.global_markdup on one
    rank must give the flags of the library's own duplicate marking."""
    import ctypes as C
    from speedseq_amd import capi, dist as sdist
    n_pairs = 400
    pairs, seqs, seq, off = common.sim_reads(n_pairs, 33, dup_frac=0.25)
    idx = emu_lib.index_load(common.EXAMPLE_FA)
    opt = emu_lib.opt_init()
    pb = np.zeros(n_pairs, dtype=np.int32)
    sig = np.zeros((n_pairs, 3), dtype=np.uint64)
    dup_local = np.zeros(n_pairs, dtype=np.uint8)
    summary = np.zeros(8, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    emu_lib._chk(emu_lib.l.ssg_hotpath_dev_sig(idx, p(opt), C.c_int(n_pairs), C.c_int(150), p(seq), p(off), p(pb), C.c_int(1), C.c_int64(0),
                                               p(summary), p(dup_local), p(sig)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29400 + os.getpid() % 500)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        s = torch.from_numpy(sig.view(np.int64).copy())
        valid = s[:, 0] != -1
        dup = sdist.global_markdup(s, valid, torch.arange(n_pairs, dtype=torch.int64))
    finally:
        dist.destroy_process_group()
    assert np.array_equal(dup.numpy(), dup_local) and dup_local.sum() > 30
    emu_lib.index_destroy(idx)


def _sort_worker(rank, world, port, d):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speedseq_amd import dist as sdist
    keys = torch.from_numpy(np.load(os.path.join(d, "keys.npy")))
    n = keys.numel()
    mine = torch.arange(rank, n, world)                 # records dealt round-robin: every rank holds keys from the whole range
    recs = torch.stack([(mine % 251).to(torch.uint8), (mine // 251 % 251).to(torch.uint8), (keys[mine] & 0xff).to(torch.uint8)], 1)
    k, o, r, sent = sdist.coordinate_range_exchange(keys[mine], mine, recs, n_samples=64)
    np.save(os.path.join(d, "out%d.npy" % rank), np.stack([k.numpy(), o.numpy()]))
    np.save(os.path.join(d, "rec%d.npy" % rank), r.numpy())
    assert sent >= 0
    dist.barrier()
    dist.destroy_process_group()


def test_coordinate_range_exchange_three_ranks_gloo():
    """SURVEY 8e coupling 3: the sample-sort merge by samtools' coordinate key leaves rank r with the r-th stretch of the globally
    sorted stream, ties in input order (stable), for keys with heavy ties and the unplaced (tid = -1: top bits set) tail."""
    rng = np.random.default_rng(5)
    n = 20000
    tid = rng.integers(0, 3, size=n).astype(np.uint64)
    tid[rng.random(n) < 0.05] = 0xffffffff                                   # unplaced lines sort last (uint64 order)
    pos = rng.integers(0, 500, size=n).astype(np.uint64)                      # many equal keys
    keys = ((tid << np.uint64(32)) | (pos << np.uint64(1)) | rng.integers(0, 2, size=n).astype(np.uint64)).view(np.int64)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "keys.npy"), keys)
        port = 31500 + os.getpid() % 2000
        mp.spawn(_sort_worker, args=(3, port, d), nprocs=3, join=True)
        outs = [np.load(os.path.join(d, "out%d.npy" % r)) for r in range(3)]
        recs = [np.load(os.path.join(d, "rec%d.npy" % r)) for r in range(3)]
    got_k = np.concatenate([o[0] for o in outs]); got_o = np.concatenate([o[1] for o in outs]); got_r = np.concatenate(recs)
    exp = np.lexsort((np.arange(n), keys.view(np.uint64)))                    # stable by unsigned key
    assert np.array_equal(got_o, exp) and np.array_equal(got_k, keys[exp])
    assert np.array_equal(got_r[:, 0], (exp % 251).astype(np.uint8)) and np.array_equal(got_r[:, 2], (keys[exp] & 0xff).astype(np.uint8))
    assert all(len(o[0]) > n // 10 for o in outs)                             # the splitters balance the ranges


def test_bench_two_rank_flow_on_the_emulation(emu_lib, tmp_path):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one JSON line from rank 0), on the host emulation with gloo in
    place of RCCL: the N > 1 step (global duplicate marking, coordinate range exchange) end to end at toy size.  Test infrastructure:
    never a number."""
    import json
    import sys
    env = dict(os.environ, SSG_EMU_DEVICES="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29583",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emu-selftest", "--cpu-sample", "0", "--partial", str(tmp_path / "p.json")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "range exchange" in d["config"]["sorted_merge"]
    m = d["literal_multi"]                              # rank 0 also runs the reference's script with bin/bwa on both (emulated) devices
    assert m.get("devices") == 2 and m.get("pairs_per_s", 0) > 0 and m.get("bai_written") is True, m
    k = d["literal_ranks"]                              # ... and as two pipelines side by side (bin/speedseq-ranks)
    assert k.get("devices") == 2 and k.get("pairs_per_s", 0) > 0 and k.get("bai_written") is True, k
    assert abs(k["bam_bytes"][".bam"] - m["bam_bytes"][".bam"]) < 0.02 * m["bam_bytes"][".bam"]   # the same records (tests/test_ranks.py compares them), BGZF blocks cut at the ranks' stretches
