# tuning aid: run the bench workload with SSG_DEBUG=2 and print the device phase counters
# (matesw [0..7], chain [8..15], chain2aln [16..23]; see the kernels for the slot meanings)
import sys, os, ctypes as C
import runpy, json, io, contextlib
os.environ["SSG_DEBUG"] = "2"
os.environ["SSGPU_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "speedseq_amd", "libssgpu_tune.so")  # make tune
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
sys.path.insert(0, '.')
from speedseq_amd import capi
lib = capi.Lib()
out = (C.c_ulonglong * 24)()
lib.l.ssg_dbg_cycles(out)
t = list(out)
print("matesw: fetch=%d sw=%d resort=%d rows=%d wave_total=%d" % tuple(t[:5]))
print("chain (lane cycles): insert=%d sort=%d weight=%d filter=%d flatten=%d | sum nc=%d sum nc^2=%d sum kept=%d" % tuple(t[8:16]))
s = max(1, sum(t[8:13]))
print("chain fractions: insert %.3f sort %.3f weight %.3f filter %.3f flatten %.3f" % tuple(x / s for x in t[8:13]))
print("chain2aln (wave cycles): window+seedsort=%d contain=%d extend=%d resort=%d wave_total=%d | chains=%d ext_seeds=%d regions=%d" % tuple(t[16:24]))
s = max(1, t[20])
print("chain2aln fractions of wave time: window %.3f contain %.3f extend %.3f resort %.3f" % tuple(x / s for x in t[16:20]))
