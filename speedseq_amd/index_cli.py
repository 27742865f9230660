"""`bwa index <ref.fa>` back-end: builds the FM-index on the GPU (index_build.py) and writes
ref.fa.{amb,ann,bwt,pac,sa} in upstream's on-disk format (reference bin/speedseq:386-391 calls
`$BWA index $REF` when any of the five files is missing).  Ns are replaced by random bases drawn like
upstream (srand48(11); lrand48() & 3) and recorded as holes in .amb."""
import ctypes
import sys

import numpy as np
import torch

from . import index_build


def read_fasta_codes(path):
    libc = ctypes.CDLL(None)
    libc.srand48(11)
    libc.lrand48.restype = ctypes.c_long
    lut = np.full(256, 4, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
        lut[ord(chr(c).lower())] = i
    names, seqs, holes = [], [], []
    name, chunks = None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip()
            if line.startswith(b">"):
                if name is not None:
                    names.append(name)
                    seqs.append(np.concatenate(chunks) if chunks else np.zeros(0, np.uint8))
                name, chunks = line[1:].split()[0].decode(), []
            else:
                chunks.append(np.frombuffer(line, dtype=np.uint8).copy())
    if name is not None:
        names.append(name)
        seqs.append(np.concatenate(chunks) if chunks else np.zeros(0, np.uint8))
    codes, off = [], 0
    for s in seqs:
        c = lut[s]
        amb = np.nonzero(c == 4)[0]
        last = None
        for p in amb:                      # upstream order: one lrand48() per ambiguous base
            ch = chr(s[p])
            if holes and last is not None and p == last + 1 and holes[-1][2] == ch:
                holes[-1][1] += 1
            else:
                holes.append([off + int(p), 1, ch])
            last = p
            c[p] = libc.lrand48() & 3
        codes.append(c)
        off += len(c)
    return names, [len(c) for c in codes], np.concatenate(codes), holes


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m speedseq_amd.index_cli <ref.fa>")
    if not torch.cuda.is_available():
        raise SystemExit("bwa index (speedseq_amd): no GPU visible; there is no CPU path")
    names, lens, codes, holes = read_fasta_codes(sys.argv[1])
    ix = index_build.build_index_arrays(torch.from_numpy(codes).cuda())
    index_build.write_index_files(sys.argv[1], ix, names, lens)
    if holes:                               # .amb with the recorded holes
        with open(sys.argv[1] + ".amb", "w") as f:
            f.write("%d %d %d\n" % (sum(lens), len(names), len(holes)))
            for o, l, ch in holes:
                f.write("%d %d %s\n" % (o, l, ch))


if __name__ == "__main__":
    main()
