#!/usr/bin/env python3
"""Seeded synthetic reference / paired-end read generator (numpy only).

Stands in for the reference's missing example FASTQ (.MISSING_LARGE_BLOBS:1) and for wgsim
(SURVEY.md section 8d): fragments ~N(d, s) sampled uniformly, substitution + indel errors, and the
injected content SURVEY 8d asks for so that samblaster paths are exercised: exact fragment
duplicates, chimeric (split) reads, and discordant (everted / long-insert) pairs.
Read names carry the truth: <prefix><i>_<contig>_<pos1>_<pos2>_<kind>.
"""
import argparse
import gzip
import numpy as np

COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
BASES = np.frombuffer(b"ACGTN", dtype=np.uint8)


def synth_reference(total_len, n_contigs=1, seed=20150810, repeat_frac=0.05):
    """Random reference with planted repeat families (so seed multiplicity is not trivially 1)."""
    rng = np.random.default_rng(seed)
    lens = np.full(n_contigs, total_len // n_contigs, dtype=np.int64)
    lens[0] += total_len - lens.sum()
    contigs = []
    fam = [rng.integers(0, 4, size=int(l), dtype=np.uint8) for l in rng.integers(200, 3000, size=8)]
    for ci, L in enumerate(lens):
        seq = rng.integers(0, 4, size=int(L), dtype=np.uint8)
        planted = 0
        while planted < repeat_frac * L:
            f = fam[rng.integers(0, len(fam))].copy()
            div = rng.random() * 0.1
            mut = rng.random(f.size) < div
            f[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
            if rng.random() < 0.5:
                f = COMP[f[::-1]]
            if f.size >= L:
                break
            p = rng.integers(0, L - f.size)
            seq[p:p + f.size] = f
            planted += f.size
        contigs.append(("chr%d" % (ci + 1), seq))
    return contigs


def read_fasta(path):
    contigs, name, chunks = [], None, []
    op = gzip.open if path.endswith(".gz") else open
    lut = np.full(256, 4, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
        lut[ord(chr(c).lower())] = i
    with op(path, "rb") as f:
        for line in f:
            line = line.rstrip()
            if line.startswith(b">"):
                if name is not None:
                    contigs.append((name, np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)))
                name, chunks = line[1:].split()[0].decode(), []
            else:
                chunks.append(lut[np.frombuffer(line, dtype=np.uint8)])
    if name is not None:
        contigs.append((name, np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)))
    return contigs


def write_fasta(path, contigs, width=60):
    with open(path, "wb") as f:
        for name, seq in contigs:
            f.write(b">" + name.encode() + b"\n")
            s = BASES[seq].tobytes()
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + b"\n")


def _mutate(rng, frag, err, indel_frac):
    """Apply substitution and indel errors to one read (uint8 codes)."""
    out = frag.copy()
    m = rng.random(out.size) < err
    if m.any():
        out[m] = (out[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) % 4
    if indel_frac > 0 and rng.random() < indel_frac * out.size:
        p = int(rng.integers(5, max(6, out.size - 5)))
        l = int(rng.integers(1, 6))
        if rng.random() < 0.5:  # insertion
            out = np.concatenate([out[:p], rng.integers(0, 4, size=l, dtype=np.uint8), out[p:]])
        else:  # deletion
            out = np.concatenate([out[:p], out[p + l:]])
    return out


def simulate(contigs, n_pairs, read_len=150, ins_mean=400, ins_std=50, err=0.005, indel_frac=0.0005,
             dup_frac=0.05, chim_frac=0.01, disc_frac=0.01, n_frac=0.001, seed=11, prefix="r"):
    """Returns list of (name, seq1 codes, seq2 codes). seq2 is already reverse-complemented (FR library)."""
    rng = np.random.default_rng(seed)
    lens = np.array([s.size for _, s in contigs], dtype=np.int64)
    prob = lens / lens.sum()
    out = []
    frags = []
    for i in range(n_pairs):
        kind = "n"
        u = rng.random()
        if frags and u < dup_frac:
            ci, pos, d = frags[int(rng.integers(0, len(frags)))]
            kind = "d"
        else:
            ci = int(rng.choice(len(contigs), p=prob))
            d = max(read_len + 10, int(rng.normal(ins_mean, ins_std)))
            if u < dup_frac + disc_frac:
                d = int(rng.integers(5000, 50000))
                kind = "x"
            L = int(lens[ci])
            d = min(d, L - 1)
            pos = int(rng.integers(0, L - d))
            frags.append((ci, pos, d))
        seq = contigs[ci][1]
        need = read_len + 8
        f1 = seq[pos:pos + need]
        f2 = COMP[seq[pos + d - need:pos + d][::-1]] if pos + d - need >= 0 else COMP[seq[0:pos + d][::-1]]
        if kind == "n" and rng.random() < chim_frac:  # chimeric read 1: prefix from here, suffix from elsewhere
            cj = int(rng.choice(len(contigs), p=prob))
            q = int(rng.integers(0, max(1, lens[cj] - need)))
            bp = int(rng.integers(40, read_len - 40))
            other = contigs[cj][1][q:q + need]
            if rng.random() < 0.5:
                other = COMP[other[::-1]]
            f1 = np.concatenate([f1[:bp], other[:need - bp]])
            kind = "c"
        if rng.random() < 0.5:  # fragment strand
            f1, f2 = f2, f1
        r1 = _mutate(rng, f1, err, indel_frac)[:read_len]
        r2 = _mutate(rng, f2, err, indel_frac)[:read_len]
        for r in (r1, r2):
            m = rng.random(r.size) < n_frac
            r[m] = 4
        out.append(("%s%d_%s_%d_%d_%s" % (prefix, i, contigs[ci][0], pos + 1, pos + d, kind), r1, r2))
    return out


def write_fastq(path, pairs, interleaved=True, path2=None):
    op = gzip.open if path.endswith(".gz") else open
    f1 = op(path, "wb")
    f2 = f1 if interleaved else (gzip.open if path2.endswith(".gz") else open)(path2, "wb")
    for name, r1, r2 in pairs:
        f1.write(b"@" + name.encode() + b"/1\n" + BASES[r1].tobytes() + b"\n+\n" + b"I" * r1.size + b"\n")
        f2.write(b"@" + name.encode() + b"/2\n" + BASES[r2].tobytes() + b"\n+\n" + b"I" * r2.size + b"\n")
    f1.close()
    if f2 is not f1:
        f2.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", help="existing FASTA; otherwise a synthetic one is generated")
    ap.add_argument("--ref-out", help="write the synthetic reference here")
    ap.add_argument("--ref-len", type=int, default=1000000)
    ap.add_argument("--contigs", type=int, default=2)
    ap.add_argument("-N", type=int, default=1000)
    ap.add_argument("-l", type=int, default=150)
    ap.add_argument("-d", type=int, default=400)
    ap.add_argument("-s", type=int, default=50)
    ap.add_argument("-e", type=float, default=0.005)
    ap.add_argument("-S", type=int, default=11)
    ap.add_argument("-o", required=True, help="interleaved FASTQ output (.gz ok)")
    a = ap.parse_args()
    if a.ref:
        contigs = read_fasta(a.ref)
    else:
        contigs = synth_reference(a.ref_len, a.contigs)
        if a.ref_out:
            write_fasta(a.ref_out, contigs)
    pairs = simulate(contigs, a.N, a.l, a.d, a.s, a.e, seed=a.S)
    write_fastq(a.o, pairs)


if __name__ == "__main__":
    main()
