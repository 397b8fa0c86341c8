"""GPU-side packing of the input letters (first half; mhip_volume_pack): the volume made on the device from the
file's bytes equals the one the host splitter makes — which tests/test_host_cpu.py pins to the reference's split_raw_dataset byte for byte —
through the C ABI and through the driver (MECAT_HIP_SPLIT=gpu)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
BIN = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2pw")


def _records(seed, n, width, letters=b"ACGT", extra=b""):
    """n reads as FASTA text with lines of `width` residues (0: one line per read) -> (text, seq_start, line_width, lens, residues per read)"""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(letters, dtype=np.uint8)
    text = bytearray()
    starts, lens, seqs = [], [], []
    for r in range(n):
        L = int(rng.integers(1, 700)) if r % 11 else int(rng.integers(1, 5))
        s = lut[rng.integers(0, len(lut), size=L)].copy()
        if extra and L > 8:
            s[rng.integers(0, L, size=max(1, L // 40))] = np.frombuffer(extra, dtype=np.uint8)[rng.integers(0, len(extra), size=max(1, L // 40))]
        text += b">r%d some text\n" % r
        starts.append(len(text))
        raw = s.tobytes()
        if width:
            text += b"\n".join(raw[i:i + width] for i in range(0, L, width)) + b"\n"
        else:
            text += raw + b"\n"
        lens.append(L)
        seqs.append(raw)
    return bytes(text), np.array(starts, dtype=np.int64), np.full(n, width, dtype=np.int32), np.array(lens, dtype=np.int32), seqs


def _host_pack(seqs):
    """PackedDB::set_char as the reference applies it (packed_db.h:98-107: the 4-bit value ORed into a 2-bit field, unmasked, inside one byte)"""
    letters, vals = "-ACMGRSVTWYHKDBN", [15, 0, 1, 6, 2, 4, 9, 13, 3, 8, 5, 12, 7, 11, 10, 14]
    enc = np.zeros(256, dtype=np.uint8)
    for ch, v in zip(letters, vals):
        enc[ord(ch)] = v
        enc[ord(ch.lower())] = v
    total = sum(len(s) + 1 for s in seqs)
    pac = np.zeros((total + 3) // 4, dtype=np.uint8)
    offs = np.zeros((len(seqs), 2), dtype=np.int32)
    pos = 0
    for r, s in enumerate(seqs):
        offs[r] = (pos, len(s))
        v = enc[np.frombuffer(s, dtype=np.uint8)].astype(np.uint16)
        idx = pos + np.arange(len(s))
        np.bitwise_or.at(pac, idx >> 2, ((v << ((~idx & 3) << 1)) & 0xFF).astype(np.uint8))
        pos += len(s) + 1
    return pac, offs, total


@pytest.mark.parametrize("width,extra", [(0, b""), (60, b""), (70, b"NRYKMSWBDHVn-acgt"), (0, b"Nn")])
def test_volume_packed_on_the_device_equals_the_host_packing(width, extra):
    import mecat_amd.hip as M
    text, starts, lws, lens, seqs = _records(5 + width, 3000, width, extra=extra)
    want, offs, nb = _host_pack(seqs)
    ctx = M.Context(0)
    vol, got = M.Volume.from_letters(ctx, text, starts, lws, offs, nb)
    assert got.tobytes() == want.tobytes()
    # and the resident volume is usable: its index equals the index of the uploaded host volume
    a = M.Index(ctx, vol)
    b = M.Index(ctx, M.Volume(ctx, want, offs, nb, 0))
    sa, oa = a.download()
    sb, ob = b.download()
    assert np.array_equal(sa, sb) and np.array_equal(oa, ob)


def test_driver_with_the_device_packer_writes_the_same_volume_files(tmp_path):
    """MECAT_HIP_SPLIT=gpu: same wrk/vol*, fileindex.txt and candidates as the host splitter, for a multi-line FASTA with IUPAC codes that
    spans two volumes (MECAT_HIP_MCS shrinks the volume for the test); the trace line says the device packed."""
    codes, lens = H.synth_reads(1200, 3000, 0.15, 400000, 12, 0)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    rng = np.random.default_rng(3)
    fa = str(tmp_path / "ml.fa")
    with open(fa, "wb") as f:
        for r in range(len(lens)):
            s = lut[codes[starts[r]:starts[r + 1]]].copy()
            if r % 9 == 0:
                s[rng.integers(0, len(s), size=3)] = np.frombuffer(b"NRy", dtype=np.uint8)
            raw = s.tobytes()
            f.write(b">%d\n" % r + b"\n".join(raw[i:i + 80] for i in range(0, len(raw), 80)) + b"\n")
    outs = {}
    for mode in ("host", "gpu"):
        w = str(tmp_path / ("w_" + mode))
        out = str(tmp_path / (mode + ".can"))
        env = dict(os.environ, MECAT_TRACE="1", MECAT_HIP_MCS="2000000")
        if mode == "gpu":
            env["MECAT_HIP_SPLIT"] = "gpu"
        r = subprocess.run([BIN, "-j", "0", "-d", fa, "-o", out, "-w", w, "-t", "4"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("pack (device)" in r.stderr) == (mode == "gpu"), r.stderr[-1500:]
        vols = sorted(x for x in os.listdir(w) if x.startswith("vol"))
        outs[mode] = ([hashlib.sha256(open(os.path.join(w, x), "rb").read()).hexdigest() for x in vols], sorted(open(out).read().splitlines()))
    assert len(outs["host"][0]) >= 2
    assert outs["host"] == outs["gpu"]


def test_driver_keeps_the_host_packer_for_records_with_ragged_lines(tmp_path):
    """a record whose lines differ in length (other than a shorter last one) is outside what mhip_volume_pack takes: under
    MECAT_HIP_SPLIT=gpu the splitter packs such a volume on the host threads, same files"""
    codes, lens = H.synth_reads(60, 2500, 0.15, 60000, 13, 0)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    fa = str(tmp_path / "ragged.fa")
    with open(fa, "wb") as f:
        for r in range(len(lens)):
            raw = lut[codes[starts[r]:starts[r + 1]]].tobytes()
            cuts = [0, 70, 130, 200] if r == 7 else list(range(0, len(raw), 60))      # read 7: lines of 70, 60, 70 ... bases
            if r == 7:
                cuts += list(range(270, len(raw), 70))
            cuts.append(len(raw))
            f.write(b">%d\n" % r + b"\n".join(raw[a:b] for a, b in zip(cuts, cuts[1:]) if b > a) + b"\n")
    outs = {}
    for mode in ("host", "gpu"):
        w = str(tmp_path / ("w_" + mode))
        out = str(tmp_path / (mode + ".can"))
        env = dict(os.environ, MECAT_TRACE="1")
        if mode == "gpu":
            env["MECAT_HIP_SPLIT"] = "gpu"
        r = subprocess.run([BIN, "-j", "0", "-d", fa, "-o", out, "-w", w, "-t", "4"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "pack (device)" not in r.stderr
        outs[mode] = (hashlib.sha256(open(os.path.join(w, "vol0"), "rb").read()).hexdigest(), sorted(open(out).read().splitlines()))
    assert outs["host"] == outs["gpu"]
