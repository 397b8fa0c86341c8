#!/usr/bin/env python3
"""Last step of a round's profile: ties the committed PMC summaries to the kernel sources they were measured on.

    python profiles/finish.py r02 [/tmp/align.s]

Writes <tag>_pmc_source.json = {"kernel_source_digest": sha256 over mecat_amd/csrc/* (bench.py: src_digest), ...} — bench.py quotes
`roofline.traffic` and the instruction counts of <tag>_hbm_traffic.json / <tag>_instruction_mix.json only while the digest equals
the one of the sources it runs on — and adds the static share of 4-cycle-class VALU instructions of dw_extend2
(tools/isa_mix.py on hipcc's assembly) to <tag>_instruction_mix.json."""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def main():
    tag = sys.argv[1]
    import bench
    meta = {"kernel_source_digest": bench.src_digest(), "command": "python bench.py --steps 1 --warmup 1 --no-cpu --no-extras --no-e2e",
            "tool": "rocprofv3 --kernel-trace --stats / --pmc (separate passes), tools/dev/profile_bench.sh"}
    # (dw_extend2ILb0 = the <false> instantiation, mecat2pw's aligner: up to round 5 the first function whose name holds "dw_extend2" was taken,
    # which is the <true> instantiation — mecat2cns' forward pass with its row log — since that one exists)
    asm = sys.argv[2] if len(sys.argv) > 2 else "/tmp/align_%s.s" % tag
    if not os.path.exists(asm):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "mecat_amd", "csrc"), "-S", "--cuda-device-only", "-x", "hip",
                        os.path.join(ROOT, "mecat_amd", "csrc", "align.hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), asm, "dw_extend2ILb0"], stdout=subprocess.PIPE, text=True, check=True).stdout
    m = re.search(r"valu 4-cycle share: ([0-9.]+)", out)
    p = os.path.join(HERE, "%s_instruction_mix.json" % tag)
    mix = json.load(open(p))
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), asm, "dw_extend2ILb0", "--row-loops"], stdout=subprocess.PIPE, text=True,
                          check=True).stdout
    m2 = re.search(r"valu 4-cycle share: ([0-9.]+)", out2)
    if m and "dw_extend2" in mix:
        mix["dw_extend2"]["valu_4cycle_fraction_static"] = float(m.group(1))
        meta["dw_extend2_static_mix"] = out.strip().splitlines()
        if m2:      # the d-row loops only (set-up, traceback and accounting code left out): what the kernel spends 92 % of its time in
            mix["dw_extend2"]["valu_4cycle_fraction_row_loops"] = float(m2.group(1))
            meta["dw_extend2_row_loop_mix"] = out2.strip().splitlines()
    json.dump(mix, open(p, "w"), indent=1, sort_keys=True)
    json.dump(meta, open(os.path.join(HERE, "%s_pmc_source.json" % tag), "w"), indent=1, sort_keys=True)
    print(json.dumps(meta)[:300])


if __name__ == "__main__":
    main()
