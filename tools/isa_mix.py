#!/usr/bin/env python3
"""isa_mix.py — static instruction-class mix of one kernel in hipcc's assembly output (-S --cuda-device-only).

Classes follow the issue costs measured by mecat_amd/bin/valu_peak on gfx950 (profiles/r02_valu_peak.json):
  valu2   wave64 VALU instructions that can issue in 2 cycles: 32-bit add/sub/and/or/xor/not/mov/lshr/ashr (and the 16-bit
          VOP2 forms), VGPR / inline-constant / literal operands only
  valu4   everything else on the VALU: min/max, shift left, multiplies, compares, v_cndmask, ffbh/bfrev, every VOP3 / VOP3P /
          DPP / SDWA encoding, any SGPR operand, readlane/readfirstlane, v_permlane*_swap (8 cycles, counted here)
  salu, lds (ds_*), vmem (global_/buffer_/flat_/scratch_), smem (s_load*), branch, waitcnt, nop, other
Also reports `vcc_rereads`: v_cndmask reads of VCC beyond the first after each write of VCC (the slow pattern: a chain of
selects on one compare through VCC issues 2-4x slower than through an SGPR pair).

    python tools/isa_mix.py build/align.s dw_extend2 [--ops] [--row-loops]

--row-loops: only the basic blocks of dw_extend2's d-row loops — the depth-2 loops (LLVM's loop annotations in the assembly) that
contain the half-wave row maximum (v_permlane16_swap), with the snake loops nested in them; set-up, tail traceback and accounting
code are left out (the share of 4-cycle-class instructions that prices the kernel must come from the loop it spends
its time in)."""
import re
import sys
from collections import Counter

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_max_i16", "v_min_i16", "v_add_u16", "v_sub_u16", "v_max_u16", "v_min_u16", "v_add_f32", "v_sub_f32",
        "v_mul_f32", "v_fma_f32", "v_fmac_f32"}


def classify(op, args):
    if op.startswith("v_"):
        base = re.sub(r"_e32$|_e64$", "", op)
        enc64 = op.endswith("_e64") or "_dpp" in op or "_sdwa" in op
        has_sgpr = bool(re.search(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|exec|vcc_lo|vcc_hi|m0)(?![a-z0-9_])", args)) and not op.startswith("v_cndmask")
        if base in FULL and not enc64 and not has_sgpr:
            return "valu2"
        return "valu4"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if re.match(r"^_Z\d+%s[A-Z]\S*:" % re.escape(name), ln) or re.match(r"^%s:" % re.escape(name), ln):
            on = True
            continue
        if on:
            if ln.strip().startswith("s_endpgm"):
                break
            out.append(ln.rstrip("\n"))
    return out


def row_loop_lines(lines):
    """lines of the blocks that belong to a depth-2 loop containing v_permlane16_swap (and to the loops nested inside it)"""
    blocks = []
    for ln in lines:
        m = re.match(r"^\.L(BB\d+_\d+):", ln)
        if m or re.match(r"^; %bb\.\d+:", ln) or not blocks:
            blocks.append({"label": m.group(1) if m else None, "lines": []})
        blocks[-1]["lines"].append(ln)
    parent, depth_of = {}, {}
    for b in blocks:
        head = "\n".join(b["lines"][:16])
        mh = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", head)
        mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", b["lines"][0])
        if mh and b["label"]:
            b["inner"] = b["label"]
            depth_of[b["label"]] = int(mh.group(1))
            ps = [(int(d), h) for h, d in re.findall(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", head)]
            parent[b["label"]] = max(ps)[1] if ps else None
        elif mi:
            b["inner"] = mi.group(1)
        else:
            b["inner"] = None

    def chain(h):
        out = []
        while h:
            out.append(h)
            h = parent.get(h)
        return out
    has_swap = set()
    for b in blocks:
        if b["inner"] and any("v_permlane16_swap" in ln for ln in b["lines"]):
            has_swap.update(chain(b["inner"]))
    rows = {h for h in has_swap if depth_of.get(h) == 2}
    keep = []
    for b in blocks:
        if b["inner"] and any(h in rows for h in chain(b["inner"])):
            keep += b["lines"]
    return keep, sorted(rows)


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, name)
    if not lines:
        sys.exit("kernel %s not found in %s" % (name, path))
    if "--row-loops" in sys.argv:
        lines, rows = row_loop_lines(lines)
        print("row loops (depth 2, with the half-wave maximum): %s" % " ".join(rows))
    mix, ops = Counter(), Counter()
    vcc_reads_since_write, vcc_rereads = 0, 0
    for ln in lines:
        s = ln.split(";")[0].strip()
        if not s or s.endswith(":") or s.startswith("."):
            if s.endswith(":"):
                vcc_reads_since_write = 0
            continue
        m = re.match(r"([a-z_0-9]+)\s*(.*)", s)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        c = classify(op, args)
        mix[c] += 1
        ops[(c, re.sub(r"_e32$|_e64$", "", op))] += 1
        dst = args.split(",")[0].strip()
        writes_vcc = dst in ("vcc", "vcc_lo") or (op.startswith(("v_add_co", "v_sub_co", "v_subrev_co")) and "vcc" in args.split(",")[1])
        if op.startswith("v_cndmask") and args.rstrip().endswith("vcc"):
            vcc_reads_since_write += 1
            if vcc_reads_since_write > 1:
                vcc_rereads += 1
        if writes_vcc:
            vcc_reads_since_write = 0
    tot = sum(mix.values())
    v2, v4 = mix["valu2"], mix["valu4"]
    print("kernel %s: %d instructions" % (name, tot))
    for k, v in mix.most_common():
        print("  %-8s %6d  %5.1f %%" % (k, v, 100.0 * v / tot))
    print("  valu 4-cycle share: %.3f   v_cndmask re-reads of VCC: %d" % (v4 / max(1, v2 + v4), vcc_rereads))
    if "--ops" in sys.argv:
        for (c, op), v in ops.most_common(60):
            print("    %-6s %-28s %5d" % (c, op, v))


if __name__ == "__main__":
    main()
