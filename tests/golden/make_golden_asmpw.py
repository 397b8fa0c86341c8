#!/usr/bin/env python3
"""Golden outputs of mecat2canu's overlappers for corrected reads: the UNMODIFIED mecat2asmpw and
mecat2trimpw (oracle/_ref/, compiled with gcc from /root/reference/mecat2canu/src/mecat2asmpw/*.c) run through their own
command line (-P<blocks dir> -T<threads> -S<start block> -E<last block>, reading <dir>/ovlprep and <dir>/00000N.fasta as canu's
Overlapmecat2asmpw.pm:483-503 lays them out) on a seeded synthetic set of corrected reads (2 % error) in two blocks.
Build container only.  Writes tests/golden/asmpw.json (+ the sorted outputs of both tools and both start blocks)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402

GEN = dict(nreads=400, L=6000, err=0.02, genome=120000, seed=91, ont=0)
BLOCKS = [(1, 200), (201, 400)]          # canu numbers reads from 1


def layout(d):
    codes, lens = H.synth_reads(GEN["nreads"], GEN["L"], GEN["err"], GEN["genome"], GEN["seed"], GEN["ont"])
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    with open(os.path.join(d, "ovlprep"), "w") as f:
        for b, e in BLOCKS:
            f.write("-allreads -allbases -b %d -e %d\n" % (b, e))
    for k, (b, e) in enumerate(BLOCKS):
        with open(os.path.join(d, "%06d.fasta" % (k + 1)), "w") as f:
            for rid in range(b, e + 1):
                s = codes[starts[rid - 1]: starts[rid]]
                f.write(">%d\n%s\n" % (rid, "".join("ACGT"[c] for c in s)))
    return lens


def run(tool, d, start, threads=2):
    exe = os.path.join(H.ROOT, "oracle", "_ref", tool)
    subprocess.run([exe, "-P" + d, "-T%d" % threads, "-S%d" % start, "-E%d" % len(BLOCKS)], check=True, stdout=subprocess.DEVNULL)
    lines = []
    for t in range(threads):
        p = os.path.join(d, "%d_%d.r" % (start, t))
        lines += open(p).read().splitlines()
        os.unlink(p)
    return sorted(lines)


def main():
    meta = {"gen": GEN, "blocks": BLOCKS, "outputs": {}}
    for tool in ("mecat2asmpw", "mecat2trimpw"):
        for start in (1, 2):
            d = tempfile.mkdtemp(prefix="asmpw_")
            layout(d)
            lines = run(tool, d, start)
            again = run(tool, d, start, threads=3)
            assert lines == again, "output depends on the thread count"
            name = "%s.S%d.sorted" % (tool, start)
            open(os.path.join(H.GOLDEN, name), "w").write("\n".join(lines) + "\n")
            meta["outputs"][name] = {"lines": len(lines), "sha256": hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest()}
            print(tool, start, len(lines), file=sys.stderr)
    json.dump(meta, open(os.path.join(H.GOLDEN, "asmpw.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
