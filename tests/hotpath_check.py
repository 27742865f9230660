"""Checker run by tests/test_gpu_kernels.py::test_gpu_hotpath_batches_and_dups (TEST INFRASTRUCTURE): the bench's step on device-resident
reads in three upstream batches -- duplicate flags equal the oracle's `bwa mem` (one insert-size model per upstream batch) +
samblaster over the whole input; the device classification's SAM / discordant / splitter line counts equal the oracle's streams."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, "tools"))
import common  # noqa: E402
import oracle_py  # noqa: E402
from speedseq_amd import capi  # noqa: E402

lib = capi.Lib()
oracle = oracle_py.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
n_pairs, per = 3000, 1000
pairs, seqs, seq, off = common.sim_reads(n_pairs, seed=41, dup_frac=0.1)
pb = (np.arange(n_pairs) // per).astype(np.int32)
gidx, oidx = lib.index_load(common.EXAMPLE_FA), oracle.idx_load(common.EXAMPLE_FA)
opt = lib.opt_init()
d_seq, d_off, d_pb = torch.from_numpy(seq).cuda(), torch.from_numpy(off).cuda(), torch.from_numpy(pb).cuda()
torch.cuda.synchronize()
summary, dup = capi.hotpath_dev(lib, gidx, opt, n_pairs, 150, d_seq.data_ptr(), d_off.data_ptr(), d_pb.data_ptr(), 3, 0, True)
s16, _ = capi.hotpath_dev_ex(lib, gidx, opt, n_pairs, 150, d_seq.data_ptr(), d_off.data_ptr(), d_pb.data_ptr(), 3, 0)
names = []
for nm, _, _ in pairs:
    names += [nm, nm]
text = ""
for b in range(3):
    lo, hi = 2 * per * b, 2 * per * (b + 1)
    t, _, _ = oracle.process_pairs(oidx, seq[off[lo]:off[hi]], off[lo:hi + 1] - off[lo], names[lo:hi], None, lo, "", 4)
    text += t
oflags, marked = common.oracle_dup_flags(oracle, text, "@SQ\tSN:20_slice\tLN:321635\n")
assert np.array_equal(dup, oflags) and oflags.sum() > 100, (int(dup.sum()), int(oflags.sum()))
assert int(s16[10]) == text.count("\n") and int(s16[1]) == int(oflags.sum()), (s16, text.count("\n"))
n_disc = sum(1 for l in oracle.last_discordants.split("\n") if l and l[0] != "@")
n_spl = sum(1 for l in oracle.last_splitters.split("\n") if l and l[0] != "@")
assert (int(s16[8]), int(s16[9])) == (n_disc, n_spl) and n_disc > 0 and n_spl > 0, (s16, n_disc, n_spl)
print("hotpath ok", int(s16[10]), int(s16[1]), n_disc, n_spl)
