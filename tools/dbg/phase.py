# tuning aid: run bench-like workload once and print the matesw phase cycle counters
import sys, os, ctypes as C, subprocess
sys.argv = [sys.argv[0]] + sys.argv[1:]
import runpy, json, io, contextlib
os.environ.setdefault("SSG_PHASE", "1")
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
sys.path.insert(0, '.')
from speedseq_amd import capi
lib = capi.Lib()
out = (C.c_ulonglong * 8)()
lib.l.ssg_dbg_cycles(out)
tot = list(out)
print("matesw phase cycles (sum over waves): fetch=%d sw=%d resort=%d rows=%d wave_total=%d" % tuple(tot[:5]))
print("fractions of wave time: fetch %.2f sw %.2f resort %.2f" % (tot[0]/tot[4], tot[1]/tot[4], tot[2]/tot[4]))
print("resort cycles by n_in: <=8: %.2f  9..64: %.2f  >64: %.2f (fractions of resort)" % (tot[5]/max(1,tot[2]), tot[6]/max(1,tot[2]), tot[7]/max(1,tot[2])))
