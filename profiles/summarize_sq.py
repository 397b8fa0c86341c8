#!/usr/bin/env python3
"""Instruction mix per kernel from rocprofv3 --pmc SQ_* passes of `python bench.py --steps 1 --warmup 1 --no-cpu --no-extras`.

    python profiles/summarize_sq.py r01 2 gpurun_out/pmc_dw1 gpurun_out/pmc_dw2 gpurun_out/pmc_dw3

Writes <tag>_instruction_mix.md / .json (per launch; argv[2] = hot-path passes per run is only used for the heading).
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    tag, passes, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    here = os.environ.get("PROFILES_OUT") or os.path.dirname(os.path.abspath(__file__))
    os.makedirs(here, exist_ok=True)
    label = os.environ.get("PROFILE_LABEL", "config 2")
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in dirs:
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                k = "xd_extend_wide" if k.startswith("xd_extend_w<true>") else k.split("(")[0].split("<")[0][:60]
                if "at::" in k or "rocclr" in k or "rocprim" in k.lower() or k.startswith("k_"):
                    continue
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]] += 1
    names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES",
             "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT",
             "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU2", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_BRANCH"]
    out = {}
    with open(os.path.join(here, "%s_instruction_mix.md" % tag), "w") as f:
        f.write("# %s - SQ instruction counters per launch (rocprofv3 --pmc, %d hot-path passes per run, %s)\n\n" % (tag, passes, label))
        f.write("Wave-level instruction counts; SQ_ACTIVE_* / SQ_WAVE_CYCLES / SQ_WAIT_* are in quad-cycles (MI355X_MICROARCH.md).\n"
                "Issue ceilings are not nominal figures here: mecat_amd/bin/valu_peak measures them on the device (profiles/*_valu_peak.json:\n"
                "a wave64 VALU instruction issues in 4 cycles per SIMD, in 2 only for the plain 32-bit add / sub / logic / mov / lshr class\n"
                "when two waves of a SIMD co-issue; SALU 4 cycles per SIMD).\n\n")
        f.write("| kernel | " + " | ".join(n.replace("SQ_", "") for n in names) + " |\n|---|" + "---|" * len(names) + "\n")
        for k in sorted(tot, key=lambda k: -tot[k].get("SQ_INSTS_VALU", 0)):
            row = []
            for n in names:
                ln = launches[k].get(n, 0)
                row.append("%.3g" % (tot[k][n] / ln) if ln else "-")
            f.write("| %s | %s |\n" % (k, " | ".join(row)))
            lv = launches[k].get("SQ_INSTS_VALU", 0)
            if lv:
                out[k] = {"valu_insts_per_launch": tot[k]["SQ_INSTS_VALU"] / lv, "salu_insts_per_launch": tot[k]["SQ_INSTS_SALU"] / lv,
                          "lds_insts_per_launch": tot[k]["SQ_INSTS_LDS"] / lv}
    json.dump(out, open(os.path.join(here, "%s_instruction_mix.json" % tag), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
