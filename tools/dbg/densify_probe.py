# tuning aid (MI355X): time of the denser suffix-array copy under SSG_DENSIFY_REFILL (lanes that wait before a wave refills), a fresh process each
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--load":
    import ctypes as C
    from speedseq_amd import capi
    lib = capi.Lib(); h = C.c_void_p()
    assert lib.l.ssg_index_load2(sys.argv[2].encode(), 0, C.byref(h)) == 0
    lib.index_destroy(h); sys.exit(0)
import numpy as np, torch
import bench
from speedseq_amd import capi
dev = torch.device("cuda", 0)
lib = capi.Lib()
ref, lens, _ = bench.synth_reference(int(3100e6), 20150810, dev)
names = bench.GRCH37_NAMES[:len(lens)]
ctg_off = np.concatenate([[0], np.cumsum(lens)])[:-1]
idx = lib.index_build_dev(ref.data_ptr(), int(ref.numel()), ctg_off, lens, names)
td = tempfile.mkdtemp(dir="/dev/shm")
prefix = os.path.join(td, "ref.fa")
lib.index_save(idx, prefix); lib.index_destroy(idx); del ref; torch.cuda.empty_cache()
for cfg in sys.argv[1:] or ["1", "8", "16", "32", "48", "64"]:
    env = dict(os.environ, SSG_LOAD_LOG="1")
    for kv in cfg.split(","):
        if "=" in kv: k, v = kv.split("=", 1); env[k] = v
        else: env["SSG_DENSIFY_REFILL"] = kv
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--load", prefix], env=env, capture_output=True, text=True)
    print(cfg, "|", " ".join(l for l in r.stderr.split("\n") if "index load" in l)[:200], r.stderr[-200:] if r.returncode else "")
import shutil; shutil.rmtree(td)
