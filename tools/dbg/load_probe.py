# tuning aid (MI355X): how fast the index files reach HBM, alone on the box -- SSG_LOAD_THREADS 2..32, a fresh process each (the files stay in the page cache)
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--load":
    from speedseq_amd import capi
    lib = capi.Lib()
    import ctypes as C
    h = C.c_void_p()
    t0 = time.time(); rc = lib.l.ssg_index_load2(sys.argv[2].encode(), 1, C.byref(h)); t1 = time.time(); idx = h
    assert rc == 0
    print("load %.3f s" % (t1 - t0)); lib.index_destroy(idx); sys.exit(0)
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
print("index files:", {e: os.path.getsize(prefix + e) for e in (".bwt", ".sa", ".pac")})
for th in (2, 4, 8, 16, 32):
    for rep in range(2):
        env = dict(os.environ, SSG_LOAD_THREADS=str(th), SSG_LOAD_LOG="1", SSG_BWA_DENSIFY_AFTER="1")
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--load", prefix], env=env, capture_output=True, text=True)
        print("threads %2d: process %.2f s | %s | %s" % (th, time.time() - t0, r.stdout.strip(), " ".join(l for l in r.stderr.split("\n") if "index load" in l)[:160]))
import shutil; shutil.rmtree(td)
