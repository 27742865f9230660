#!/usr/bin/env python3
"""An alarm, not a freeze: the machine code of every ssg_k_* kernel of libssgpu.so (gfx950, all translation units) against the build that
last passed `pytest -m gpu` and the bench's parity gate on an MI355X (tests/golden/kernel_isa.sha256).

There is no GPU where the CPU-side suite runs, and this toolchain has compiled source that is right under the host emulation into code
that is wrong on the GPU (a `continue` out of the middle of the seeding kernel's extension site; round 4 found the
statement on the MI355X).  A kernel whose code differs from the pin -- because its source changed, or because something else in its
translation unit did: adding a kernel re-schedules its neighbours -- has not been on a GPU yet; the test says so.  The GPU scripts
(tools/gpu_r*.sh) write the new pin themselves after the suite and the parity gate are green (`--write --golden gpurun_out/...`), and the file is
copied into tests/golden/ with the commit that ships those kernels.
usage: isa_pin.py [--write] [--golden FILE] [--lib speedseq_amd/libssgpu.so]"""
import hashlib
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
GOLDEN = os.path.join(ROOT, "tests", "golden", "kernel_isa.sha256")


def code_objects(lib):
    """gfx950 code objects embedded in the shared library (one clang offload bundle per translation unit)"""
    data = open(lib, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if i < 0:
            return out
        n, = struct.unpack_from("<Q", data, i + 24)
        o = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            trip = data[o:o + tl].decode()
            o += tl
            if "gfx950" in trip and size:
                out.append(data[i + off:i + off + size])
        pos = i + 24


def kernel_hashes(co_bytes, tmp="/tmp/isa_pin.co"):
    open(tmp, "wb").write(co_bytes)
    txt = subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", "--no-leading-addr", tmp], text=True)
    cur, body, res = None, [], {}
    pcrel = re.compile(r"^(s_addc?_u32 s\d+, s\d+, )0x[0-9a-f]+$")         # offsets to constant data move with the rest of the unit

    def flush():
        if cur and cur.startswith("_Z") and "ssg_k_" in cur or (cur or "").startswith("ssg_k_"):
            res[cur] = hashlib.sha256("\n".join(body).encode()).hexdigest()
    for l in txt.split("\n"):
        m = re.match(r"^<(.*)>:$", l)
        if m:
            flush()
            cur, body = m.group(1), []
            continue
        l = re.sub(r"//.*", "", l)
        l = re.sub(r"\s+", " ", l).strip()
        if l:
            body.append(pcrel.sub(r"\1PCREL", l))
    flush()
    return res


def current(lib):
    cos = code_objects(lib)
    if not cos:
        raise SystemExit("no gfx950 code object in %s" % lib)
    res = {}
    for c in cos:                                                        # one code object per translation unit (core, index build, seeding, table)
        res.update(kernel_hashes(c))
    return res


def main():
    lib = os.path.join(ROOT, "speedseq_amd", "libssgpu.so")
    if "--lib" in sys.argv:
        lib = sys.argv[sys.argv.index("--lib") + 1]
    h = current(lib)
    golden = sys.argv[sys.argv.index("--golden") + 1] if "--golden" in sys.argv else GOLDEN   # the GPU scripts write the pin themselves once the suite and the parity gate are green
    if "--write" in sys.argv:
        with open(golden, "w") as f:
            for k in sorted(h):
                f.write("%s  %s\n" % (h[k], k))
        print("pinned %d kernels" % len(h))
        return 0
    want = dict((l.split("  ", 1)[1].strip(), l.split("  ", 1)[0]) for l in open(golden) if l.strip())
    bad = [k for k in want if h.get(k) != want[k]]
    new = [k for k in h if k not in want]
    for k in bad:
        print("CHANGED" if k in h else "MISSING", k[:150])
    for k in new:
        print("NEW (not pinned)", k[:150])
    print("%d pinned kernels, %d changed or missing, %d new" % (len(want), len(bad), len(new)))
    return 1 if bad or new else 0


if __name__ == "__main__":
    sys.exit(main())
