# tuning aid: chain_wave phase counters (cycles of lane 0, summed over reads of the 2048 / 4096 classes) on the bench workload; needs `make tune`
import sys, os, ctypes as C, runpy, json, io, contextlib
os.environ["SSGPU_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "speedseq_amd", "libssgpu_tune.so")
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-e2e", "--cpu-sample", "0"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().split("\n")[-1])
k = d["roofline"]["kernels_ms_per_step"]
print(d["ms_per_step"], {x: k[x] for x in k if "chain" in x})
sys.path.insert(0, ".")
from speedseq_amd import capi
lib = capi.Lib()
out = (C.c_ulonglong * 32)()
lib.l.ssg_dbg_cycles(out)
t = list(out)
n = max(1, t[12])
print("reads %d  seeds/read %.0f  chains/read %.0f" % (t[12], t[13] / n, t[14] / n))
print("cycles per read: insert %.0f  weights %.0f  introsort %.0f  filter %.0f" % (t[8] / n, t[9] / n, t[10] / n, t[11] / n))
