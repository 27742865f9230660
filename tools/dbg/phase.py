# tuning aid: run the bench workload with SSG_DEBUG=2 and print the device phase counters
# (matesw [0..7], chain [8..15], chain2aln [16..23]; see the kernels for the slot meanings)
import sys, os, ctypes as C
import runpy, json, io, contextlib
os.environ["SSG_DEBUG"] = "2"
os.environ["SSGPU_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "speedseq_amd", "libssgpu_tune.so")  # make tune
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-e2e", "--cpu-sample", "0"]   # the timed step only
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
sys.path.insert(0, '.')
from speedseq_amd import capi
lib = capi.Lib()
out = (C.c_ulonglong * 32)()
lib.l.ssg_dbg_cycles(out)
t = list(out)
print("matesw: fetch=%d sw=%d resort=%d rows=%d wave_total=%d" % tuple(t[:5]))
print("pair_final (lane cycles): mark_primary=%d mem_pair+mapq=%d no_pairing=%d xa=%d | sum (n0+n1)^2=%d" % tuple(t[8:13]))
s = max(1, sum(t[8:12]))
print("pair_final fractions: mark_primary %.3f mem_pair %.3f no_pairing %.3f xa %.3f" % tuple(x / s for x in t[8:12]))
print("matesw resort by n_in: <=8 %.2f  9..64 %.2f  >64 %.2f (of resort); resort/total %.2f sw/total %.2f" % (t[5]/max(1,t[2]), t[6]/max(1,t[2]), t[7]/max(1,t[2]), t[2]/max(1,t[4]), t[1]/max(1,t[4])))
print("chain2aln (wave cycles): chain record+seed order=%d containment scan=%d (unused)=%d re-sort=%d wave_total=%d | scan chunks=%d append=%d results+build=%d" % tuple(t[16:24]))
s = max(1, t[20])
print("chain2aln fractions of wave time: record %.3f scan %.3f resort %.3f append %.3f results+build %.3f; cycles per scan chunk %.0f" % (t[16]/s, t[17]/s, t[19]/s, t[22]/s, t[23]/s, t[17]/max(1,t[21])))
r = max(1, t[26])
print("smem (wave cycles): state machine=%d extension site=%d | rounds=%d, ready lanes per round %.1f of %.1f alive" % (t[24], t[25], t[26], 64.0 * t[27] / r, 64.0 * t[28] / r))
o2 = (C.c_ulonglong * 16)()
lib.l.ssg_dbg_cycles_at(64, 16, o2)
u = list(o2)
tot = max(1, sum(u[:7]))
print("sort_dedup_fast (lane-0 cycles, chain2aln + matesw): keys %.3f rank sort(re) %.3f introsort(re ties) %.3f scan %.3f rank sort(score) %.3f introsort(score ties) %.3f gather %.3f of %d"
      % tuple([x / tot for x in u[:7]] + [tot]))
print("  calls %d (regions %d, sum n^2 %d); with ties in re: %d calls, %d regions" % (u[10], u[11], u[12], u[8], u[9]))
