#!/usr/bin/env python3
"""tests/golden/dense.json: the UNMODIFIED reference (oracle/_ref/mecat2pw, built with the reference's own release flags) on the deep,
repeat-rich set of helpers.dense_reads() with `-n 1500 -k 2` — the only way to make reads keep more than 1 000 overlaps under the
index's bucket cap of 128 — so that the per-read m4 sort (pw_impl.cpp:581) works on lists of >= 1000 records.
Build container only (about six minutes on 32 threads):  python tests/golden/make_golden_dense.py"""
import collections
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402

codes, lens = H.dense_reads()
d = tempfile.mkdtemp(prefix="dense_")
fa = os.path.join(d, "dense.fa")
H.write_fasta(fa, codes, lens)
out = {"reads": int(len(lens)), "bases": int(lens.sum()), "fasta_sha256": H.sha256_lines(open(fa).read().splitlines())}
for name, args, col in (("can", ["-j", "0"], 0), ("m4_g1", ["-j", "1", "-g", "1"], 1)):
    o = os.path.join(d, name)
    subprocess.run([H.ref_bin(), "-d", fa, "-o", o, "-w", os.path.join(d, "w_" + name), "-t", "32", "-n", "1500", "-k", "2"] + args, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = sorted(open(o).read().splitlines())
    per = collections.Counter(l.split()[col] for l in lines)
    out[name] = {"args": args + ["-n", "1500", "-k", "2"], "lines": len(lines), "sorted_sha256": H.sha256_lines(lines),
                 "most_lines_of_one_query_read": max(per.values()), "query_reads_with_1000_lines_or_more": sum(1 for v in per.values() if v >= 1000)}
    print(name, out[name], flush=True)
json.dump(out, open(os.path.join(H.GOLDEN, "dense.json"), "w"), indent=1)
