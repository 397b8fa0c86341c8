#!/usr/bin/env python3
"""Golden outputs of the 50-candidate variants of mecat2canu's overlappers (mecat2asmpw50 / mecat2trimpw50: MAXC 50,
mecat2canu/src/mecat2asmpw/mecat2asmpw50.c:23) — the UNMODIFIED tools (oracle/_ref/, gcc on the reference's files) on a read set dense
enough that reads have more than 50 candidates against a block, so the top-MAXC cut (mecat2asmpw.c:711-722) decides which candidates
survive.  The tools' worker threads keep the seeds of the reads they mapped before (divergence table), which can move a
candidate's score by a few votes: the set is run with one, two and three threads, and what differs between those runs — output the
reference itself does not reproduce — is recorded beside the lines every run agrees on.
Build container only.  Writes tests/golden/asmpw50.json + the sorted -T1 outputs."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402

GEN = dict(nreads=600, L=6000, err=0.02, genome=45000, seed=92, ont=0)      # 80x: ~75 overlapping reads per read and block
BLOCKS = [(1, 300), (301, 600)]


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


def run(tool, d, start, threads):
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
    for tool in ("mecat2asmpw50", "mecat2trimpw50"):
        for start in (1, 2):
            d = tempfile.mkdtemp(prefix="asmpw50_")
            layout(d)
            runs = {t: run(tool, d, start, t) for t in (1, 2, 3)}
            common = set(runs[1]) & set(runs[2]) & set(runs[3])
            name = "%s.S%d.sorted" % (tool, start)
            open(os.path.join(H.GOLDEN, name), "w").write("\n".join(runs[1]) + "\n")
            per_read = {}
            for ln in runs[1]:
                per_read[ln.split("\t")[0]] = per_read.get(ln.split("\t")[0], 0) + 1
            meta["outputs"][name] = {"lines": len(runs[1]), "sha256": hashlib.sha256(("\n".join(runs[1]) + "\n").encode()).hexdigest(),
                                     "lines_T2": len(runs[2]), "lines_T3": len(runs[3]), "lines_in_every_run": len(common),
                                     "thread_dependent_lines": sorted(set(runs[1]) ^ common | set(runs[2]) ^ common | set(runs[3]) ^ common)[:200],
                                     "max_lines_per_read": max(per_read.values()) if per_read else 0}
            print(tool, start, {k: v for k, v in meta["outputs"][name].items() if k != "thread_dependent_lines"},
                  "thread-dependent:", len(meta["outputs"][name]["thread_dependent_lines"]), file=sys.stderr)
    json.dump(meta, open(os.path.join(H.GOLDEN, "asmpw50.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
