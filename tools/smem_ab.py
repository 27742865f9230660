#!/usr/bin/env python3
"""A/B timing of the seeding stage inside the bench's device step (MI355X only): one synthetic reference, one index, one batch of reads,
then the timed step under several configurations -- environment switches of the product library and / or variant libraries
(speedseq_amd/libssgpu_<name>.so, `make variant`) -- with the per-kernel HIP-event times, the step time and the step's summary counts
(which must not depend on the configuration).

usage: smem_ab.py [--pairs N] [--ref-mbp M] [--steps K] CONFIG...
  CONFIG = name[@lib][:VAR=value[,VAR=value...]]      e.g.  quad:SSG_SMEM_KERNEL=quad   b2000:SSG_SMEM_MAX_EXT=2000   w2@seed_w2
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from speedseq_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1000000)
    ap.add_argument("--ref-mbp", type=float, default=3100.0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--kernels", default="smem", help="comma-separated substrings of the kernel names whose times are listed per configuration")
    ap.add_argument("--dbg-cycles", action="store_true", help="with an instrumented library (make tune): print the device phase counters after every configuration")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "smem_ab.json"))
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib0 = capi.Lib(None)
    opt = lib0.opt_init()
    t0 = time.time()
    ref, lens, _ = bench.synth_reference(int(a.ref_mbp * 1e6), 20150810, dev)
    torch.cuda.synchronize()
    ctg_off = np.concatenate([[0], np.cumsum(lens)])[:-1]
    names = bench.GRCH37_NAMES[:len(lens)]
    idx0 = lib0.index_build_dev(ref.data_ptr(), int(ref.numel()), ctg_off, lens, names)
    bench.log("reference + index: %.1f s" % (time.time() - t0))
    rl = a.read_len
    reads = bench.simulate_pairs(ref, lens, a.pairs, rl, 12, dev, ins_mean=800 if rl >= 250 else 400, ins_std=150 if rl >= 250 else 50)
    d_seq = reads.reshape(-1)
    d_off = (torch.arange(2 * a.pairs + 1, device=dev, dtype=torch.int64) * rl).contiguous()
    pb, n_batches = bench.bwa_batches(a.pairs, rl, 16)
    d_pb = torch.from_numpy(pb).to(dev)
    del ref
    torch.cuda.empty_cache()
    libs = {None: (lib0, idx0)}
    prefix = None
    results = []
    for cfg in a.configs:
        head, _, envs = cfg.partition(":")
        name, _, libname = head.partition("@")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        if (libname or None) not in libs:
            if prefix is None:
                prefix = "/tmp/smem_ab_idx"
                t0 = time.time(); lib0.index_save(idx0, prefix); bench.log("index saved for the variant libraries: %.1f s" % (time.time() - t0))
            l = capi.Lib(os.path.join(ROOT, "speedseq_amd", "libssgpu_%s.so" % libname))
            l._chk(l.l.ssg_set_device(C.c_int(0)))
            libs[libname] = (l, l.index_load(prefix))
        lib, idx = libs[libname or None]
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            # two switches are read when an index is made, not per step: the table of short-pattern intervals is rebuilt for this configuration (and
            # again, at its default, after it); the suffix-array samples only ever get denser, so such configurations go last, in falling order
            if "SSG_KTAB_K" in env:
                lib._chk(lib.l.ssg_index_build_ktab(idx))
            if "SSG_SA_INTV" in env:
                lib._chk(lib.l.ssg_index_densify_to(idx, C.c_int(int(env["SSG_SA_INTV"]))))
            capi.hotpath_dev_ex(lib, idx, opt, a.pairs, rl, d_seq.data_ptr(), d_off.data_ptr(), d_pb.data_ptr(), n_batches, 0)   # warm-up
            torch.cuda.synchronize()
            lib.l.ssg_prof_reset(); lib.l.ssg_prof_enable(C.c_int(1))
            t0 = time.perf_counter()
            for _ in range(a.steps):
                summary, _ = capi.hotpath_dev_ex(lib, idx, opt, a.pairs, rl, d_seq.data_ptr(), d_off.data_ptr(), d_pb.data_ptr(), n_batches, 0)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            kern = capi.prof_get(lib)
            lib.l.ssg_prof_enable(C.c_int(0))
            seed_ms = {k: round(v[0] / a.steps, 2) for k, v in kern.items() if any(x in k for x in a.kernels.split(","))}
            dbg = None
            if a.dbg_cycles:
                out = (C.c_ulonglong * 32)()
                lib.l.ssg_dbg_cycles(out)
                dbg = [int(x) for x in out]
            r = {"config": cfg, "dbg_cycles": dbg, "ms_per_step": round(1e3 * dt, 1), "seeding_kernels_ms": seed_ms, "records": int(summary[0]), "seeds": int(summary[2]), "bwt_extends": int(summary[6]),
                 "chains": int(summary[7]), "dup_pairs": int(summary[1]), "sam_lines": int(summary[10])}
        except Exception as e:   # a configuration that fails must not take the others with it
            r = {"config": cfg, "error": repr(e)}
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            if "SSG_KTAB_K" in env:
                lib.l.ssg_index_build_ktab(idx)
        results.append(r)
        bench.log(json.dumps(r))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(results, open(a.out, "w"), indent=1)
    base = results[0]
    same = all(all(r.get(k) == base.get(k) for k in ("records", "seeds", "chains", "dup_pairs", "sam_lines")) for r in results if "error" not in r)
    print("summary counts equal over the configurations:", same)
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
