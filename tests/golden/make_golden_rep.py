#!/usr/bin/env python3
"""tests/golden/rep.json: the UNMODIFIED reference binary (oracle/_ref/mecat2pw) on the repeat-structured sets of helpers.REP_SETS and on
the config-1-sized repeat-rich set REP_CLI (interspersed families at 0 - 5 % divergence, microsatellites, homopolymer runs — buckets
at and beyond the cap of 128, the 41st-seed rule on non-self hits, tied scores).  Sorted-output hashes + line
counts per task, plus what the sets exercise (bucket and insert_loc statistics from the oracle).
Build container only:  python tests/golden/make_golden_rep.py"""
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402

out = {}
sets = dict(H.REP_SETS)
sets["rep_cli"] = H.REP_CLI
for name, (n, L, err, G, seed, ont, nfam, mc, nsat) in sorted(sets.items()):
    codes, lens, st = H.synth_reads_rep(n, L, err, G, seed, ont, nfam, mc, nsat)
    d = tempfile.mkdtemp(prefix=name + "_")
    fa = os.path.join(d, name + ".fa")
    H.write_fasta(fa, codes, lens)
    e = {"gen": dict(nreads=n, L=L, err=err, genome=G, seed=seed, ont=ont, nfam=nfam, max_copies=mc, nsat=nsat), "genome_repeats": st,
         "reads": int(len(lens)), "bases": int(lens.sum()), "fasta_sha256": H.sha256_lines(open(fa).read().splitlines()),
         "buckets": H.bucket_stats(codes, lens)}
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    H.orc_stats_reset()
    cands = H.orc_seed_all(ov, ov, oidx, H.orc_params(tech=ont))
    e["insert_loc"] = H.orc_stats()
    e["candidates"] = int(sum(len(c) for c in cands))
    for task, args in (("can", ["-j", "0"]), ("m4_g0", ["-j", "1", "-g", "0"]), ("m4_g1", ["-j", "1", "-g", "1"])):
        o = os.path.join(d, task)
        a = args + ["-x", str(ont)]
        subprocess.run([H.ref_bin(), "-d", fa, "-o", o, "-w", os.path.join(d, "w_" + task), "-t", "8"] + a, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = sorted(open(o).read().splitlines())
        e[task] = {"args": a, "lines": len(lines), "sorted_sha256": H.sha256_lines(lines)}
        if name == "rep_ont" and task == "m4_g1":
            # the reference needs 95 s for this run (one chunk of 500 reads = one thread): the CPU test compares the oracle with these lines — the
            # first 40 query reads' (column 2) — instead of starting the binary again
            with open(os.path.join(H.GOLDEN, "rep_ont.m4_g1.q40.sorted"), "w") as f:
                f.write("".join(ln + "\n" for ln in lines if int(ln.split()[1]) < 40))
    print(name, json.dumps(e), flush=True)
    out[name] = e
json.dump(out, open(os.path.join(H.GOLDEN, "rep.json"), "w"), indent=1)
