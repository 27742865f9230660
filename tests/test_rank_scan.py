"""Rank 0's input scanner (speedseq_amd/host/ranksplit.h): the several-thread scanner of plain FASTQ files must return what the one-thread scanner returns -- the same
records (start, end, sequence length), the same end, the same refusal in the same words -- whatever the slice size and thread count, on well-formed files and on the
shapes the one-thread scanner accepts or refuses by rule (no final newline, blank lines, qualities that begin with '@' or '+', CR LF, several lines per sequence,
truncation)."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "emu", "scan_test")


def _rec(rng, i, lo=30, hi=300, evil=False):
    n = rng.randint(lo, hi)
    s = "".join(rng.choice("ACGTN") for _ in range(n))
    q = "".join(rng.choice("@+IJ#5") if evil else rng.choice("FGHIJ") for _ in range(n))
    if evil and n > 2:
        q = rng.choice("@+") + q[1:]
    plus = "+" if rng.random() < 0.8 else "+read%d comment" % i
    return "@r%d%s\n%s\n%s\n%s\n" % (i, " some comment" if rng.random() < 0.3 else "", s, plus, q)


def _files(tmp_path):
    rng = random.Random(7)
    out = {}
    good = "".join(_rec(rng, i) for i in range(400))
    evil = "".join(_rec(rng, i, evil=True) for i in range(400))
    out["good"] = good
    out["evil_quals"] = evil
    out["no_final_newline"] = good[:-1]
    out["blank_lines_at_end"] = good + "\n\n  \n"
    out["blank_line_inside"] = "".join(_rec(rng, i) for i in range(150)) + "\n" + "".join(_rec(rng, i) for i in range(150, 300))
    out["crlf_in_the_middle"] = "".join(_rec(rng, i) for i in range(200)) + "@x\r\nACGT\r\n+\r\nIIII\r\n" + "".join(_rec(rng, i) for i in range(200, 260))
    out["two_line_sequence"] = "".join(_rec(rng, i) for i in range(220)) + "@m\nACGT\nACGT\n+\nIIIIIIII\n" + "".join(_rec(rng, i) for i in range(220, 260))
    out["short_quality"] = "".join(_rec(rng, i) for i in range(210)) + "@s\nACGTACGT\n+\nIIII\n" + "".join(_rec(rng, i) for i in range(210, 260))
    out["empty_sequence"] = "".join(_rec(rng, i) for i in range(190)) + "@e\n\n+\n\n" + "".join(_rec(rng, i) for i in range(190, 230))
    out["truncated"] = good[:len(good) * 2 // 3]
    out["one_record"] = _rec(rng, 0)
    out["long_records"] = "".join(_rec(rng, i, lo=3000, hi=9000) for i in range(40))
    out["not_fastq"] = ">fa\nACGT\n" * 50
    paths = {}
    for k, v in out.items():
        p = str(tmp_path / (k + ".fq"))
        open(p, "w").write(v)
        paths[k] = p
    return paths


@pytest.mark.parametrize("slice_threads", [("700", "3"), ("4096", "8"), ("50000", "2"), ("33554432", "4")])
def test_several_thread_scanner_equals_the_one_thread_scanner(tmp_path, slice_threads):
    subprocess.check_call(["make", "-s", "-C", ROOT, "tests/emu/scan_test"])   # always: make's dependency check is cheap, a stale binary tests nothing
    env = dict(os.environ, SSG_RANKS_SCAN_SLICE=slice_threads[0], SSG_RANKS_SCAN_THREADS=slice_threads[1])
    for name, path in _files(tmp_path).items():
        r = subprocess.run([EXE, path], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (name, r.stderr[-300:])
        lines = r.stdout.split("\n")
        k = [i for i, l in enumerate(lines) if l.startswith("one thread:")][0]
        one, one_sum = lines[:k], lines[k][len("one thread:"):]
        many, many_sum = lines[k + 1:-2], lines[-2][len("several threads:"):]
        assert lines[-2].startswith("several threads:"), (name, lines[-3:])
        assert one == many, name
        assert one_sum == many_sum, (name, one_sum, many_sum)
        if name in ("good", "evil_quals", "no_final_newline", "blank_lines_at_end", "blank_line_inside", "long_records"):
            assert " rc 0 " in one_sum and "records %d " % {"long_records": 40, "blank_line_inside": 300}.get(name, 400) in one_sum, (name, one_sum)
        if name in ("crlf_in_the_middle", "two_line_sequence", "short_quality", "empty_sequence", "not_fastq"):
            assert " rc -1 " in one_sum, (name, one_sum)
