#!/usr/bin/env python3
"""dev helper (GPU box): mecat_amd/bin/<tool> against the UNMODIFIED tool (oracle/_ref/<tool>, built in the container by oracle/Makefile) on
a block layout larger than the golden one — same machine, same input, sorted outputs compared line by line, both timed.
    python tools/dev/asmpw_scale.py [nreads=20000] [L=8000] [genome=5000000] [blocks=2] [tool=mecat2asmpw] [threads=32] [start=1]
ASMPW_SCALE_NOREF=1: the device tool only."""
import json, os, sys, tempfile
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
    d = tempfile.mkdtemp(prefix="asmpw_scale_")
    blocks, bases = W.asm_blocks_layout(d, n, L, G, nb, 77)
    res = {"reads": n, "bases": bases, "blocks": blocks, "tool": tool, "threads": T, "start": start}
    outs = {}
    legs = (("reference", os.path.join(ROOT, "oracle", "_ref", tool)), ("device", os.path.join(ROOT, "mecat_amd", "bin", tool)))
    if os.environ.get("ASMPW_SCALE_NOREF"):
        legs = legs[1:]
    for name, exe in legs:
        lines, secs, err = W.asm_tool_run(exe, d, T, start, nb, env=dict(os.environ, MECAT_ASMPW_TIMES="1"))
        res[name + "_seconds"] = secs
        res[name + "_lines"] = len(lines)
        outs[name] = lines
        if name == "device":
            res["device_times"] = [ln for ln in err.splitlines() if ln.startswith("[mecat2asmpw]")][-1:]
    if len(outs) == 2:
        A, B = set(outs["reference"]), set(outs["device"])
        res["identical"] = outs["reference"] == outs["device"]
        res["only_reference"] = len(A - B)
        res["only_device"] = len(B - A)
        res["examples"] = [sorted(A - B)[:3], sorted(B - A)[:3]]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
