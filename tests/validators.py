"""Oracle-INDEPENDENT checks of aligner output (TEST INFRASTRUCTURE).  Every (a)-row parity test compares the HIP path with
oracle/, and oracle/ is a recall of upstream bwa / samblaster: if the recall is wrong in one operator both sides are wrong
together.  These validators use neither: they re-derive what a SAM record claims from the reference bases (.pac), the read,
and the simulator's planted truth.
  * md_nm_consistency   POS + CIGAR + MD rebuild exactly the reference bases under the alignment; NM = mismatches + gap bases
  * as_from_cigar       the alignment score re-computed from CIGAR + bases sits in [AS - pen_clip5 - pen_clip3, AS]
                        (AS is the extension's best local score; a to-end extension may give up < pen_clip per side)
  * mate_symmetry       RNEXT / PNEXT / TLEN / 0x20 / 0x8 mirror the mate's primary line; 0x40 / 0x80; 0x1 everywhere
  * truth_recall        confidently mapped primaries (MAPQ >= 20) of non-chimeric reads lie at the simulated position
  * planted_duplicates  the pairs samblaster marks are the simulator's re-emitted fragments (except the first occurrence)
"""
import re

import numpy as np

CIG = re.compile(r"(\d+)([MIDNSHP=X])")
CODE = {c: i for i, c in enumerate("ACGTN")}


def pac_contigs(prefix):
    """reference bases as the index holds them (2-bit .pac; ambiguous bases already replaced): {name: uint8 codes}"""
    ann = open(prefix + ".ann").read().split("\n")
    l_pac, n_seq = int(ann[0].split()[0]), int(ann[0].split()[1])
    pac = np.fromfile(prefix + ".pac", dtype=np.uint8)
    codes = np.empty(pac.size * 4, dtype=np.uint8)
    for k in range(4):
        codes[k::4] = (pac >> (6 - 2 * k)) & 3
    codes = codes[:l_pac]
    out = {}
    for i in range(n_seq):
        name = ann[1 + 2 * i].split()[1]
        off, ln = int(ann[2 + 2 * i].split()[0]), int(ann[2 + 2 * i].split()[1])
        out[name] = codes[off:off + ln]
    return out


def records(sam_text):
    for line in sam_text.split("\n"):
        if line and line[0] != "@":
            f = line.split("\t")
            tags = {t[:2]: t[5:] for t in f[11:]}
            yield f, int(f[1]), tags


def _ref_from_md(read, cigar, md):
    """reference bases (as letters) under the alignment, from the read, the CIGAR and the MD string alone"""
    ops = [(int(n), op) for n, op in CIG.findall(cigar)]
    # read bases aligned by M, in order; deletions come from MD
    q, aligned = 0, []
    for n, op in ops:
        if op in "M=X":
            aligned.append(read[q:q + n]); q += n
        elif op in "IS":
            q += n
    aligned = "".join(aligned)
    ref, ai, mism, dels = [], 0, 0, 0
    for tok in re.findall(r"\d+|\^[A-Z]+|[A-Z]", md):
        if tok.isdigit():
            n = int(tok); ref.append(aligned[ai:ai + n]); ai += n
        elif tok[0] == "^":
            ref.append(tok[1:]); dels += len(tok) - 1
        else:
            ref.append(tok); ai += 1; mism += 1
    return "".join(ref), mism, dels, ai == len(aligned)


def md_nm_consistency(sam_text, contigs):
    n = bad = 0
    for f, flag, tags in records(sam_text):
        if flag & 4 or f[5] == "*" or f[9] == "*":
            continue
        ops = [(int(k), op) for k, op in CIG.findall(f[5])]
        if any(op == "H" for _, op in ops):
            read = f[9]                       # hard-clipped supplementary: SEQ already lacks the clipped bases
            ops = [(k, op) for k, op in ops if op != "H"]
            cigar = "".join("%d%s" % x for x in ops)
        else:
            read, cigar = f[9], f[5]
        ref, mism, dels, used_all = _ref_from_md(read, cigar, tags["MD"])
        ins = sum(k for k, op in ops if op == "I")
        pos = int(f[3]) - 1
        truth = "".join("ACGT"[c] for c in contigs[f[2]][pos:pos + len(ref)])
        ok = used_all and ref == truth and int(tags["NM"]) == mism + dels + ins and len(ref) == sum(k for k, op in ops if op in "MD=X")
        n += 1; bad += not ok
    return n, bad


def as_from_cigar(sam_text, contigs, a=1, b=4, o=6, e=1, clip=5):
    n = bad = 0
    for f, flag, tags in records(sam_text):
        if flag & 4 or f[5] == "*" or f[9] == "*" or "AS" not in tags:
            continue
        ops = [(int(k), op) for k, op in CIG.findall(f[5]) if op != "H"]
        read, q, r, sc = f[9], 0, int(f[3]) - 1, 0
        ref = contigs[f[2]]
        for k, op in ops:
            if op == "M":
                rb = ref[r:r + k]
                qb = np.array([CODE.get(c, 4) for c in read[q:q + k]], dtype=np.uint8)
                sc += int(np.where(qb > 3, -1, np.where(qb == rb, a, -b)).sum()); q += k; r += k
            elif op == "I":
                sc -= o + e * k; q += k
            elif op == "D":
                sc -= o + e * k; r += k
            elif op == "S":
                q += k
        AS = int(tags["AS"])
        n += 1; bad += not (AS - 2 * clip <= sc <= AS)
    return n, bad


def mate_symmetry(sam_text):
    n = bad = 0
    blocks, cur, name = [], [], None
    for f, flag, tags in records(sam_text):
        if f[0] != name:
            if cur:
                blocks.append(cur)
            cur, name = [], f[0]
        cur.append((f, flag))
    if cur:
        blocks.append(cur)
    for blk in blocks:
        prim = {fl & 0xc0: (f, fl) for f, fl in blk if not fl & 0x900}
        if 0x40 not in prim or 0x80 not in prim:
            bad += 1; n += 1; continue
        for me, mate in ((0x40, 0x80), (0x80, 0x40)):
            (f, fl), (m, mfl) = prim[me], prim[mate]
            ok = fl & 1 and bool(fl & 0x20) == bool(mfl & 0x10) and bool(fl & 0x8) == bool(mfl & 0x4)
            if not mfl & 4 or not fl & 4:     # a placed pair: coordinates mirror
                rn = m[2] if f[6] != "=" else f[2]
                ok = ok and (f[6] == "=" and f[2] == m[2] or f[6] == m[2]) and f[7] == m[3] and int(f[8]) == -int(m[8])
            n += 1; bad += not ok
        for f, fl in blk:                      # supplementary lines carry the same mate fields as their primary
            n += 1; bad += not (fl & 1 and fl & 0xc0 in (0x40, 0x80))
    return n, bad


def truth_recall(sam_text, tol=20, min_mapq=20):
    """names are <prefix><i>_<contig>_<pos1>_<pos2>_<kind> (tools/simreads.py); contig names may contain '_'"""
    n = hit = 0
    for f, flag, tags in records(sam_text):
        if flag & 0x904 or int(f[4]) < min_mapq:
            continue
        parts = f[0].split("_")
        kind, p2, p1 = parts[-1], int(parts[-2]), int(parts[-3])
        ctg = "_".join(parts[1:-3])
        if kind in "cb":
            continue                          # chimeric read 1 / a read over a contig boundary: either locus is a legitimate primary
        ref_len = sum(int(k) for k, op in CIG.findall(f[5]) if op in "MD")
        pos = int(f[3])
        lclip = int(CIG.findall(f[5])[0][0]) if CIG.findall(f[5])[0][1] in "SH" else 0
        left = abs(pos - lclip - p1) <= tol                                   # the fragment's left end
        right = abs(pos + ref_len - 1 - p2) <= tol + 8 or abs(pos - lclip + 150 - 1 - p2) <= tol + 8   # ... or its right end (reverse read)
        n += 1; hit += f[2] == ctg and (left or right)
    return n, hit


def planted_duplicates(sam_text):
    """(pairs marked, re-emitted fragments among the marked, re-emitted fragments not marked although both ends mapped).
    A pair is a re-emitted fragment when an earlier pair of the stream carries the same (contig, pos1, pos2) in its name."""
    marked = planted_marked = planted_missed = 0
    seen = set()
    for f, flag, tags in records(sam_text):
        if flag & 0x900 or not flag & 0x40:
            continue
        frag = tuple(f[0].split("_")[1:-1])
        again = frag in seen
        seen.add(frag)
        if flag & 0x400:
            marked += 1; planted_marked += again
        elif again and not flag & 0xc:
            planted_missed += 1
    return marked, planted_marked, planted_missed
