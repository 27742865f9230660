# GPU diagnosis of the fused script leg at scale: run it with a short limit and, if it has not finished, say which process / thread is waiting where
import os, signal, subprocess, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from speedseq_amd import capi
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
torch.cuda.init(); dev = torch.device("cuda", 0)
lib = capi.Lib()
ref, lens, _ = bench.synth_reference(3100000000, 20150810, dev)
idx = lib.index_build_dev(ref.data_ptr(), int(ref.numel()), np.concatenate([[0], np.cumsum(lens)])[:-1], lens, bench.GRCH37_NAMES)
td = tempfile.TemporaryDirectory(dir="/dev/shm"); prefix = os.path.join(td.name, "ref.fa")
lib.index_save(idx, prefix)
reads = bench.simulate_pairs(ref, lens, n_pairs, 150, 5, dev).cpu().numpy()
fq = os.path.join(td.name, "r.fq"); bench.write_fastq(fq, reads, 150)
lib.index_destroy(idx); del ref, reads; torch.cuda.empty_cache()
bench.log("inputs ready")
import threading
def watch():
    time.sleep(float(os.environ.get("DIAG_AFTER", "30")))
    out = subprocess.run("ps -eLo pid,tid,stat,pcpu,wchan:28,etimes,comm,args --sort=pid | grep -E 'bwa|samblaster|sambamba|awk|parallel' | grep -v grep | cut -c1-220", shell=True, capture_output=True, text=True).stdout
    lines = out.split("\n")
    sys.stderr.write("---- threads of the pipeline after the wait (%d lines; busiest and a sample) ----\n" % len(lines))
    import collections
    agg = collections.Counter()
    for l in lines:
        f = l.split()
        if len(f) > 7:
            agg[(f[0], f[6], f[4], f[2])] += 1
    for k, v in sorted(agg.items()):
        sys.stderr.write("pid %s %-12s wchan %-28s stat %-4s x%d\n" % (k[0], k[1], k[2], k[3], v))
    sys.stderr.write(subprocess.run("ls -la %s/script_diag/out* %s/script_diag/*/ 2>/dev/null | head -30" % (td.name, td.name), shell=True, capture_output=True, text=True).stdout)
threading.Thread(target=watch, daemon=True).start()
cfg = "export SSG_FUSED=1\nexport SSG_SORT_THREADS=128\nexport SSG_SORT_LOG=1\n"
b = lambda n: os.path.join(ROOT, "bin", n)
r = bench.script_leg(td.name, "diag", prefix, fq, n_pairs, 32, b("bwa"), b("samblaster"), b("sambamba"), config_extra=cfg, limit_s=int(os.environ.get("DIAG_LIMIT", "40")))
print({k: v for k, v in r.items() if k != "out"})
