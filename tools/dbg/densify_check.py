# GPU check at the headline index size: the SA samples the index load densifies by LF walks (ssg_k_sa_densify_walk) against upstream's bwt_sa,
# through the product executable on index files written by ssg_index_save (SSG_SA_VERIFY=1 prints the count of differing entries)
import os, subprocess, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from speedseq_amd import capi
torch.cuda.init(); dev = torch.device("cuda", 0)
lib = capi.Lib()
ref, lens, _ = bench.synth_reference(int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 3100000000, 20150810, dev)
idx = lib.index_build_dev(ref.data_ptr(), int(ref.numel()), np.concatenate([[0], np.cumsum(lens)])[:-1], lens, bench.GRCH37_NAMES)
td = tempfile.TemporaryDirectory(dir="/dev/shm"); prefix = os.path.join(td.name, "ref.fa")
lib.index_save(idx, prefix)
reads = bench.simulate_pairs(ref, lens, 2000, 150, 5, dev).cpu().numpy()
fq = os.path.join(td.name, "r.fq"); bench.write_fastq(fq, reads, 150)
lib.index_destroy(idx); del ref; torch.cuda.empty_cache()
for env in ({"SSG_SA_VERIFY": "1", "SSG_DEBUG": "1"},):
    r = subprocess.run([os.path.join(ROOT, "bin", "bwa"), "mem", "-p", prefix, fq], capture_output=True, text=True, env=dict(os.environ, **env))
    print("rc", r.returncode, "SAM lines", r.stdout.count("\n"))
    print("\n".join(l for l in r.stderr.split("\n") if "SA samples" in l or "index load" in l))
