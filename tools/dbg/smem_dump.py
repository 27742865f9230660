#!/usr/bin/env python3
"""Seeding intervals of the library named by SSGPU_LIB (default: the product build) against the oracle's mem_collect_intv on simulated reads
of the bundled reference: how many reads differ, and for the first few the two lists side by side (start, end, occurrences) -- which pass
the extra or missing intervals come from is visible from their shape (pass 1: SMEMs; pass 2: inside a long SMEM, more occurrences than
it; pass 3: min_seed_len-ish seeds with fewer than max_mem_intv occurrences).  Used on the GPU box with SSGPU_LIB naming a variant library (`make variant`).
usage: smem_dump.py [n_pairs] [--emu]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import common  # noqa: E402
from speedseq_amd import capi  # noqa: E402


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 500
    lib = capi.Lib(os.path.join(ROOT, "tests", "emu", "libssgpu_emu.so") if "--emu" in sys.argv else None)
    import oracle_py
    orc = oracle_py.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    oidx, gidx = orc.idx_load(common.EXAMPLE_FA), lib.index_load(common.EXAMPLE_FA)
    _, seqs, seq, off = common.sim_reads(n_pairs, 14, 150)
    intv, cnt = lib.smem_batch(gidx, lib.opt_init(), seq, off, cap=96)
    bad, tot_g, tot_o, shown = 0, 0, 0, 0
    for r, s in enumerate(seqs):
        o = orc.collect_intv(oidx, s)
        g = intv[r, :max(cnt[r], 0)]
        tot_g += int(cnt[r]); tot_o += len(o)
        if len(o) == cnt[r] and np.array_equal(o, g):
            continue
        bad += 1
        if shown < 5:
            shown += 1
            fmt = lambda a: " ".join("[%d,%d)x%d" % (int(v["info"]) >> 32, int(v["info"]) & 0xffffffff, int(v["x2"])) for v in a)
            print("read %d: library %d intervals, oracle %d\n  lib: %s\n  orc: %s" % (r, cnt[r], len(o), fmt(g), fmt(o)))
    print("%s: %d of %d reads differ; intervals: library %d, oracle %d (x %.3f)" % (os.environ.get("SSGPU_LIB", "product build"), bad, len(seqs), tot_g, tot_o, tot_g / max(1, tot_o)))
    lib.index_destroy(gidx)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
