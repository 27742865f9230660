# tuning aid (MI355X): does the device slow down under sustained load?  The bench step, over and over for about two minutes, with the clocks and the power rocm-smi reports.
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from speedseq_amd import capi
def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in o.split("\n") if any(k in l for k in ("sclk", "mclk", "Power", "Temperature (Sensor junction)", "Temperature (Sensor memory)", "Performance Level"))]
        return " | ".join(keep)[:600]
    except Exception as e:
        return "rocm-smi: %r" % e
dev = torch.device("cuda", 0)
lib = capi.Lib()
ref, lens, _ = bench.synth_reference(int(3100e6), 20150810, dev)
names = bench.GRCH37_NAMES[:len(lens)]
ctg_off = np.concatenate([[0], np.cumsum(lens)])[:-1]
idx = lib.index_build_dev(ref.data_ptr(), int(ref.numel()), ctg_off, lens, names)
n = 1000000; rl = 150
reads = bench.simulate_pairs(ref, lens, n, rl, 12, dev)
d_seq = reads.reshape(-1); d_off = (torch.arange(2 * n + 1, device=dev, dtype=torch.int64) * rl).contiguous()
pb, nb = bench.bwa_batches(n, rl, 16); d_pb = torch.from_numpy(pb).to(dev)
del ref; torch.cuda.empty_cache()
opt = lib.opt_init()
print("idle:", smi())
t_begin = time.time(); k = 0
while time.time() - t_begin < float(sys.argv[1] if len(sys.argv) > 1 else 120):
    t0 = time.time()
    capi.hotpath_dev(lib, idx, opt, n, rl, d_seq.data_ptr(), d_off.data_ptr(), d_pb.data_ptr(), nb, 0)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if k % 20 == 0:
        print("t=%5.1f s step %3d: %.1f ms | %s" % (time.time() - t_begin, k, dt * 1e3, smi()))
    k += 1
print("after:", smi())
