# tuning aid: lane utilisation of the lane-per-extension kernel (k_extlane.h, `make tune`) on the bench workload.
# Rows of a wave's lanes run in step, so a wave pays, per row index, the trips of its widest lane, for as many rows as its longest lane.
import sys, os, ctypes as C
import runpy, json, io, contextlib
os.environ["SSGPU_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "speedseq_amd", "libssgpu_tune.so")
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-e2e", "--cpu-sample", "0", "--no-profile", "--config5-pairs", "0", "--no-dist-rehearsal"] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
sys.path.insert(0, '.')
from speedseq_amd import capi
lib = capi.Lib()
out = (C.c_ulonglong * 32)()
lib.l.ssg_dbg_cycles_hi(out)
t = list(out)
for name, b in (("U=2 (sides <= 72)", 0), ("U=4 (longer sides)", 8)):
    lt, lr, ln, wt, wr, wn = t[b:b + 6]
    if not wn:
        continue
    print("%s: %d lanes in %d waves (%.1f per wave); trips: lanes %d, waves x 64 %d -> lane utilisation %.3f; rows: per lane %.1f, per wave %.1f (row utilisation %.3f); trips per row: lane %.2f, wave (widest) %.2f"
          % (name, ln, wn, ln / wn, lt, 64 * wt, lt / max(1, 64 * wt), lr / max(1, ln), wr / wn, lr / max(1, 64 * wr), lt / max(1, lr), wt / max(1, wr)))
print("requests listed for the DP kernel by band width w2 (0: lengths differ only; <=4, <=8, <=12, <=16, <=24, <=32, <=48, <=64, <100, >=100; last: not computable):", t[16:27], t[31])
