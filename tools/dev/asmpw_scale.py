#!/usr/bin/env python3
"""dev helper (GPU box): mecat_amd/bin/<tool> against the UNMODIFIED tool (oracle/_ref/<tool>, built in the container by oracle/Makefile) on
a block layout larger than the golden one — same machine, same input, sorted outputs compared line by line, both timed.
    python tools/dev/asmpw_scale.py [nreads=20000] [L=8000] [genome=5000000] [blocks=2] [tool=mecat2asmpw] [threads=32] [start=1]"""
import os, subprocess, sys, tempfile, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mecat_amd import workload as W

def main():
    a = sys.argv[1:]
    n = int(a[0]) if len(a) > 0 else 20000
    L = int(a[1]) if len(a) > 1 else 8000
    G = int(a[2]) if len(a) > 2 else 5_000_000
    nb = int(a[3]) if len(a) > 3 else 2
    tool = a[4] if len(a) > 4 else "mecat2asmpw"
    T = int(a[5]) if len(a) > 5 else 32
    start = int(a[6]) if len(a) > 6 else 1
    codes, lens = W.synth_reads(n, L, 0.02, G, 77, 0)
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    d = tempfile.mkdtemp(prefix="asmpw_scale_")
    per = (n + nb - 1) // nb
    blocks = [(k * per + 1, min(n, (k + 1) * per)) for k in range(nb)]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(os.path.join(d, "ovlprep"), "w") as f:
        for b, e in blocks:
            f.write("-allreads -allbases -b %d -e %d\n" % (b, e))
    for k, (b, e) in enumerate(blocks):
        with open(os.path.join(d, "%06d.fasta" % (k + 1)), "wb") as f:
            for rid in range(b, e + 1):
                f.write(b">%d\n" % rid + lut[codes[starts[rid - 1]: starts[rid]]].tobytes() + b"\n")
    res = {"reads": n, "bases": int(lens.sum()), "blocks": blocks, "tool": tool, "threads": T, "start": start}
    outs = {}
    legs = (("reference", os.path.join(ROOT, "oracle", "_ref", tool)), ("device", os.path.join(ROOT, "mecat_amd", "bin", tool)))
    if os.environ.get("ASMPW_SCALE_NOREF"):
        legs = legs[1:]
    for name, exe in legs:
        t0 = time.time()
        r = subprocess.run([exe, "-P" + d, "-T%d" % T, "-S%d" % start, "-E%d" % nb], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, MECAT_ASMPW_TIMES="1"))
        res[name + "_seconds"] = time.time() - t0
        if name == "device":
            res["device_times"] = [ln for ln in r.stderr.splitlines() if ln.startswith("[mecat2asmpw]")][-1:]
        if r.returncode != 0:
            res[name + "_error"] = r.stderr[-500:]
            break
        lines = []
        for t in range(T):
            p = os.path.join(d, "%d_%d.r" % (start, t))
            lines += open(p).read().splitlines()
            os.unlink(p)
        lines.sort()
        outs[name] = lines
        res[name + "_lines"] = len(lines)
    if len(outs) == 2:
        A, B = set(outs["reference"]), set(outs["device"])
        res["identical"] = outs["reference"] == outs["device"]
        res["only_reference"] = len(A - B)
        res["only_device"] = len(B - A)
        res["examples"] = [sorted(A - B)[:3], sorted(B - A)[:3]]
    print(json.dumps(res))

if __name__ == "__main__":
    main()
