"""Checker run by tests/test_gpu_kernels.py::test_gpu_hotpath_batches_and_dups (TEST INFRASTRUCTURE): common.check_hotpath on the MI355X with
the reads in torch device tensors, in its own process because torch must initialise its HIP runtime before libssgpu is loaded (as in
bench.py).  argv[1] = read length (250: BASELINE.json config 5, 2x250 with long inserts)."""
import os
import sys

import torch

torch.cuda.init()
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, "tools"))
import common  # noqa: E402
import oracle_py  # noqa: E402
from speedseq_amd import capi  # noqa: E402

RL = int(sys.argv[1]) if len(sys.argv) > 1 else 150
lib = capi.Lib()
oracle = oracle_py.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))


def to_dev(a):
    t = torch.from_numpy(a).cuda()
    return t, t.data_ptr()


r = common.check_hotpath(lib, oracle, 3000, 1000, RL, to_dev)
torch.cuda.synchronize()
print("hotpath ok", *r)
