#!/usr/bin/env python3
"""What the host of the GPU box gives a process: visible CPUs, cgroup quota, and how zlib level 6 on BAM-shaped 64 KB blocks (the sort's
deflate stage) scales with the number of worker processes.  The literal metric's tail is host work; this says how much host there is."""
import multiprocessing as mp
import os
import random
import struct
import time
import zlib


def bam_like(n, seed):
    rng = random.Random(seed)
    recs = []
    for _ in range(n):
        name = ("r%d" % rng.randrange(10 ** 7)).encode() + b"\0"
        core = struct.pack("<iiIIiiii", rng.randrange(25), rng.randrange(10 ** 8), 0x12345678, (99 << 16) | 1, 150, rng.randrange(25), rng.randrange(10 ** 8), rng.randrange(-500, 500))
        body = core + name + struct.pack("<I", 150 << 4) + bytes(rng.getrandbits(8) for _ in range(75)) + bytes([40]) * 150 + b"NMC\x00MDZ150\x00ASC\x96XSC\x00RGZbench\x00MCZ150M\x00MQC\x3c"
        recs.append(struct.pack("<I", len(body)) + body)
    return b"".join(recs)


def work(args):
    data, secs = args
    t0 = time.time(); done = 0
    while time.time() - t0 < secs:
        for k in range(0, len(data), 65280):
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 8)
            c.compress(data[k:k + 65280]); c.flush()
        done += len(data)
    return done / (time.time() - t0)


def main():
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "loadavg", open("/proc/loadavg").read().strip())
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/memory.max"):
        if os.path.exists(f):
            print(f, open(f).read().strip())
    model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
    print("cpu model", model[0] if model else "?", "x", len(model))
    mem = [l for l in open("/proc/meminfo") if l.startswith(("MemTotal", "MemAvailable"))]
    print(" ".join(x.strip() for x in mem))
    data = bam_like(8000, 1)
    for n in (1, 8, 16, 32, 64, 128, 256):
        if n > 2 * (os.cpu_count() or 1):
            break
        with mp.Pool(n) as pool:
            r = pool.map(work, [(data, 1.5)] * n)
        print("zlib level 6, %3d processes: %7.1f MB/s total, %5.1f MB/s each" % (n, sum(r) / 1e6, sum(r) / n / 1e6))


if __name__ == "__main__":
    main()
