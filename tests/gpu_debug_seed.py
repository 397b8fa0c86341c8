#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP seeding pipeline with the oracle's seeding state (debug aid, GPU box).
usage: python tests/gpu_debug_seed.py [set=tiny] [nreads_to_check=40]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H  # noqa: E402
import mecat_amd.hip as M  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
ncheck = int(sys.argv[2]) if len(sys.argv) > 2 else 40
G = json.load(open(os.path.join(H.GOLDEN, "golden.json")))
g = G["sets"][name]["gen"]
codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g["ont"])
ov = H.orc_pack(codes, lens)
oidx = H.orc().orc_index_build(ov)
offs, pac = H.vol_arrays(ov)
ctx = M.Context(0)
gv = M.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
gidx = M.Index(ctx, gv)
L = M.lib()
L.mhip_debug_set_flags.argtypes = [C.c_void_p, C.c_int]
L.mhip_debug_strand.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int]
tech = g["ont"]
p = M.default_params(tech)
L.mhip_debug_set_flags(ctx.h, 1)
n = min(ncheck, len(lens))
M.seed_reads(ctx, gidx, gv, gv, 0, n, p)
nbits = max(1, int(ov.contents.num_bases // 2000).bit_length())
npass = (nbits + 7) // 8
in_b = npass & 1


def dbg(strand, what, dtype, count):
    a = np.zeros(max(count, 1), dtype=dtype)
    rc = L.mhip_debug_strand(ctx.h, strand, what, a.ctypes.data, a.nbytes, in_b)
    assert rc == 0, L.mhip_last_error()
    return a[:count]


O = H.orc()
bk = O.orc_bk_new(ov.contents.num_bases)
CAP = 200000
seg_ids = np.zeros(CAP, np.int32); idxs = np.zeros(CAP, np.int16); scores = np.zeros(CAP, np.int16)
lz = np.zeros(CAP * 40, np.int16); sn = np.zeros(CAP * 40, np.int16)
gate = 2 * p.min_kmer_match
nbad = 0
for rid in range(n):
    rsize = int(lens[rid])
    r1 = np.zeros(rsize + 16, np.int8)
    O.orc_extract_one_seq(ov, rid, r1.ctypes.data)
    r2 = np.zeros(rsize + 16, np.int8)
    r2[:rsize] = 3 - r1[:rsize][::-1]
    for st, rd in enumerate((r1, r2)):
        s = 2 * rid + st
        used = O.orc_seeding_state(rd.ctypes.data, rsize, oidx, bk, CAP, seg_ids.ctypes.data, idxs.ctypes.data, scores.ctypes.data, lz.ctypes.data, sn.ctypes.data)
        hdr = dbg(s, 0, np.uint32, 4)
        Hh, nseg, nrec, ng = [int(x) for x in hdr]
        keys = dbg(s, 7, np.uint64, Hh)
        problems = []
        if Hh and not np.all(keys[:-1] >> 27 <= keys[1:] >> 27):
            problems.append("keys not sorted by seg")
        gseg = dbg(s, 1, np.uint32, nseg).astype(np.int64)
        gscore = dbg(s, 2, np.int32, nseg)
        gstart = dbg(s, 3, np.uint32, nseg)
        ent = dbg(s, 4, np.uint32, nrec)
        fin = dbg(s, 5, np.uint32, nrec)
        gated = dbg(s, 6, np.uint32, ng)
        if nseg != used:
            problems.append("nseg %d != oracle used_segs %d" % (nseg, used))
        order = np.argsort(seg_ids[:used], kind="stable")
        osorted = seg_ids[:used][order]
        if nseg == used and not np.array_equal(osorted, gseg):
            problems.append("segment id sets differ")
        if not problems:
            for gi in range(nseg):
                oi = order[gi]
                sc = int(gscore[gi]) & ~0x40000000
                ovf = bool(int(gscore[gi]) & 0x40000000)
                if sc != int(scores[oi]):
                    problems.append("seg %d score gpu %d orc %d (ovf %d)" % (gseg[gi], sc, scores[oi], ovf))
                    continue
                m = min(sc, 40)
                src = fin if ovf else ent
                e = src[gstart[gi]: gstart[gi] + m]
                gl = (e >> 16).astype(np.int16); gs = (e & 0xFFFF).astype(np.uint16).astype(np.int16)
                if not (np.array_equal(gl, lz[oi * 40: oi * 40 + m]) and np.array_equal(gs, sn[oi * 40: oi * 40 + m])):
                    problems.append("seg %d lists differ (score %d ovf %d)\n gpu loc %s\n orc loc %s\n gpu seed %s\n orc seed %s" % (
                        gseg[gi], sc, ovf, gl, lz[oi * 40: oi * 40 + m], gs, sn[oi * 40: oi * 40 + m]))
            want_gated = [int(seg_ids[i]) for i in range(used) if idxs[i] >= gate]
            got_gated = [int(gseg[x]) for x in gated]
            if want_gated != got_gated:
                problems.append("gated order differs: gpu %s... orc %s..." % (got_gated[:12], want_gated[:12]))
        if problems:
            nbad += 1
            if nbad <= 6:
                print("strand", s, "H", Hh, "nseg", nseg, "nrec", nrec, "ngated", ng)
                for q in problems[:5]:
                    print("   ", q)
print("debug_seed: %d strands checked, %d with problems" % (2 * n, nbad))
