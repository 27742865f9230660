import sys, os, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import numpy as np
import common, oracle_py
from speedseq_amd import capi
lib = capi.Lib(); orc = oracle_py.Oracle('oracle/liboracle.so')
n = int(sys.argv[1])
gidx = lib.index_load(common.EXAMPLE_FA)
_, seqs, seq, off = common.sim_reads(n, 15)
t0=time.time(); ro, regs, st = lib.align1_batch(gidx, lib.opt_init(), seq, off); print("align1", n, "pairs", time.time()-t0, "s", st[:2], flush=True)
