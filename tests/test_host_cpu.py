"""CPU-side checks of the product's host logic (no GPU): the C ABI library loads and exports every symbol declared in
include/mecat_hip.h, fails loudly without a device, and the mecat2pw driver keeps the reference's CLI/file protocol."""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import helpers as H

G = json.load(open(os.path.join(H.GOLDEN, "golden.json")))
BIN = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2pw")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-s", "hip", "host", "synth"], cwd=H.ROOT, check=True, stdout=subprocess.DEVNULL)


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(H.ROOT, "include", "mecat_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(mhip_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 20
    lib = C.CDLL(os.path.join(H.ROOT, "mecat_amd", "lib", "libmecat_hip.so"))
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.mhip_abi_version() == 1


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import mecat_amd.hip as M
    with pytest.raises(M.MhipError) as e:
        M.Context(0)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_reference_oracle():
    """rule: nothing under mecat_amd/ or include/ may import, link or call oracle/"""
    bad = []
    for base in ("mecat_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(H.ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle/|liboracle|orc_[a-z]+\(|import helpers", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_cli_usage_and_validation(tmp_path):
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 1 and "usage:" in r.stderr and "dataset must be specified." in r.stderr
    r = subprocess.run([BIN, "-d", "x.fa", "-o", "o", "-w", str(tmp_path / "w"), "-j", "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "task (-j) must be 0 or 1, not 3." in r.stderr
    r = subprocess.run([BIN, "-d", "x.fa", "-o", "o", "-w", str(tmp_path / "w"), "-g", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "must be either '0' or '1'" in r.stderr
    r = subprocess.run([BIN, "-d", "x.fa", "-o", "o", "-w", str(tmp_path / "w"), "-q"], capture_output=True, text=True)
    assert r.returncode == 1 and "unrecognised option" in r.stderr


def _run_split(tmp_path, text, name="in.fa"):
    fa = tmp_path / name
    fa.write_bytes(text)
    wrk = tmp_path / ("w_" + name)
    r = subprocess.run([BIN, "-j", "0", "-d", str(fa), "-o", str(tmp_path / "o"), "-w", str(wrk)], capture_output=True, text=True)
    return r, wrk


def test_cli_volume_bytes_match_reference_golden(tmp_path):
    """FASTA -> wrk/vol0 + fileindex.txt byte-identical to the reference (the split runs before any GPU call)"""
    g = G["sets"]["tiny"]["gen"]
    codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g["ont"])
    fa = str(tmp_path / "tiny.fa")
    H.write_fasta(fa, codes, lens)
    wrk = str(tmp_path / "wrk")
    subprocess.run([BIN, "-j", "0", "-d", fa, "-o", str(tmp_path / "o"), "-w", wrk], capture_output=True)
    assert hashlib.sha256(open(os.path.join(wrk, "vol0"), "rb").read()).hexdigest() == G["sets"]["tiny"]["vol0_sha256"]
    assert open(os.path.join(wrk, "fileindex.txt")).read() == wrk + "/vol0\n"


def test_cli_fasta_fastq_quirks(tmp_path):
    """record grammar of the reference reader: FASTQ, CR/LF flavours, comments, ';' tails, lower case, multi-line"""
    O = H.orc()
    reads = ["ACGTACGTTTGACCA", "ggcattacgatcagg", "TTTTACGTACGGGTA"]
    variants = {
        "plain.fa": b">a\nACGTACGTTTGACCA\n>b\nggcattacgatcagg\n>c\nTTTTACGTACGGGTA\n",
        "crlf.fa": b">a\r\nACGTACGTTTGACCA\r\n>b\r\nggcattacgatcagg\r\n>c\r\nTTTTACGTACGGGTA",
        "cr.fa": b">a\rACGTACGTTTGACCA\r>b\rggcattacgatcagg\r>c\rTTTTACGTACGGGTA\r",
        "multi.fa": b"#comment\n>a\nACGTACG\nTTTGACCA\n\n>b\nggcattac;ignored\ngatcagg\n!x\n>c\nTTTTACGTACGGGTA\n",
        "fq.fq": b"@a\nACGTACGTTTGACCA\n+\n@@@@IIIIIIIIIII\n@b\nggcattacgatcagg\n+b\n>>>>IIIIIIIIIII\n@c\nTTTTACGTACGGGTA\n+\nIIIIIIIIIIIIIII\n",
    }
    codes = np.array([O.orc_encode_base(ord(ch)) for r in reads for ch in r], dtype=np.uint8)
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    ov = H.orc_pack(codes, lens)
    want = str(tmp_path / "want_vol0")
    O.orc_volume_dump(ov, want.encode())
    for name, text in variants.items():
        r, wrk = _run_split(tmp_path, text, name)
        assert open(os.path.join(wrk, "vol0"), "rb").read() == open(want, "rb").read(), name


def test_cli_rejects_malformed_input(tmp_path):
    r, _ = _run_split(tmp_path, b"ACGT\n>a\nACGT\n", "nodef.fa")
    assert r.returncode != 0 and "doesn't start with a defline" in r.stderr
    r, _ = _run_split(tmp_path, b">a\n>b\nACGT\n", "noseq.fa")
    assert r.returncode != 0 and "sequence data is missing" in r.stderr
    r, _ = _run_split(tmp_path, b">a\nACGT@ACGT\n", "bad.fa")
    assert r.returncode != 0 and "invalid residue" in r.stderr


def test_cli_threaded_reader_equals_sequential_reader(tmp_path):
    """plain FASTA goes through the threaded mmap reader; its volumes must equal the sequential (reference-grammar) reader's:
    ragged multi-line records, ambiguity codes (unmasked OR quirk), lower case, no final newline, several
    volumes, more threads than records (a file with empty lines always takes the sequential reader, which owns the reference's
    rules for them: test_cli_blank_line_rules_of_the_reference_reader)"""
    rng = np.random.default_rng(7)
    recs = []
    for i in range(137):
        n = int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTacgtNnRYKM-"), size=n, p=[.2, .2, .2, .2, .03, .03, .03, .03, .02, .01, .01, .01, .01, .01, .01]))
        w = int(rng.integers(1, 80))
        lines = [seq[j:j + w] for j in range(0, n, w)]
        recs.append(">r%d some text > with a bracket\n" % i + "\n".join(lines))
    text = ("\n".join(recs)).encode()            # no trailing newline
    fa = tmp_path / "ragged.fa"
    fa.write_bytes(text)
    outs = {}
    for tag, env, threads in (("seq", {"MECAT_HIP_SPLIT": "seq"}, "1"), ("t1", {}, "1"), ("t7", {}, "7"), ("t64", {}, "64")):
        wrk = tmp_path / ("w_" + tag)
        e = dict(os.environ, MECAT_HIP_MCS="3000", **env)
        r = subprocess.run([BIN, "-j", "0", "-d", str(fa), "-o", str(tmp_path / "o"), "-w", str(wrk), "-t", threads], capture_output=True,
                           text=True, env=e)
        assert "split '" in r.stderr, r.stderr
        names = [ln.strip() for ln in open(wrk / "fileindex.txt")]
        outs[tag] = (r.stderr.split("split '")[1].split("\n")[0].split("(")[1], [open(n, "rb").read() for n in names])
        assert len(names) > 5
    for tag in ("t1", "t7", "t64"):
        assert outs[tag][0] == outs["seq"][0], tag            # "(N reads, M nucls) into V volumes."
        assert outs[tag][1] == outs["seq"][1], tag


def test_cli_blank_line_rules_of_the_reference_reader(tmp_path):
    """The reference's line reader reports an EMPTY line as end-of-input once the file's last (partial) 8 MB buffer is loaded
    (buffer_line_iterator.cpp:22-73, 118-133).  Consequences that the splitter reproduces: a blank line between records is
    harmless; a blank line inside a record ends the record, so the data line after it aborts with 'doesn't start with a
    defline'; two blank lines in a row end the input (later records are dropped).  Checked against the reference binary when
    it was built here, and against the expected volumes otherwise."""
    O = H.orc()

    def vol_of(reads):
        codes = np.array([O.orc_encode_base(ord(ch)) for r in reads for ch in r], dtype=np.uint8)
        lens = np.array([len(r) for r in reads], dtype=np.int32)
        p = str(tmp_path / ("want_%d" % len(reads)))
        O.orc_volume_dump(H.orc_pack(codes, lens), p.encode())
        return open(p, "rb").read()

    a, b, c = "ACGTACGTTTGACCA", "GGCATTACGATCAGG", "TTTTACGTACGGGTA"
    cases = {
        "between.fa": (b">a\n" + a.encode() + b"\n\n>b\n" + b.encode() + b"\n\n>c\n" + c.encode() + b"\n", [a, b, c], 0),
        "trailing.fa": (b">a\n" + a.encode() + b"\n>b\n" + b.encode() + b"\n\n\n\n", [a, b], 0),
        "double.fa": (b">a\n" + a.encode() + b"\n\n\n>b\n" + b.encode() + b"\n>c\n" + c.encode() + b"\n", [a], 0),
        "inrecord.fa": (b">a\n" + a[:7].encode() + b"\n\n" + a[7:].encode() + b"\n>b\n" + b.encode() + b"\n", None, 1),
        "afterhdr.fa": (b">a\n\n" + a.encode() + b"\n", None, 1),
    }
    for name, (text, reads, fails) in cases.items():
        r, wrk = _run_split(tmp_path, text, name)
        if fails:
            assert r.returncode != 0 and ("doesn't start with a defline" in r.stderr or "sequence data is missing" in r.stderr), (name, r.stderr)
        else:
            assert open(os.path.join(wrk, "vol0"), "rb").read() == vol_of(reads), name
        if H.ref_bin():
            fa = str(tmp_path / name)
            rw = str(tmp_path / ("rw_" + name))
            rr = subprocess.run([H.ref_bin(), "-j", "0", "-d", fa, "-o", str(tmp_path / "ro"), "-w", rw, "-t", "1"], capture_output=True, text=True)
            if fails:
                assert rr.returncode != 0, name
            else:
                assert open(os.path.join(rw, "vol0"), "rb").read() == open(os.path.join(wrk, "vol0"), "rb").read(), name


def test_bench_has_no_function_local_import_of_a_module_level_name():
    """a function-local `import x` makes x local to the whole function: a use before that statement (another branch of main()) then fails
    with UnboundLocalError — which is how the N > 1 path of bench.py broke once"""
    import ast
    t = ast.parse(open(os.path.join(H.ROOT, "bench.py")).read())
    top = set()
    for n in t.body:
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            top.update(a.asname or a.name.split(".")[0] for a in n.names)
    for f in (n for n in t.body if isinstance(n, ast.FunctionDef)):
        local = set()
        for n in ast.walk(f):
            if isinstance(n, (ast.Import, ast.ImportFrom)):
                local.update(a.asname or a.name.split(".")[0] for a in n.names)
        assert not (local & top), (f.name, sorted(local & top))


def test_bench_volumes_are_the_splitter_s_volumes(tmp_path):
    """mecat_amd.workload.synth_volumes (what bench.py --workload config3 | config5_cell | config5 uploads) cuts and packs a read set exactly
    as the drop-in's splitter does (which the goldens pin to the reference's, split_database.cpp:221-266): same volume files, byte for byte,
    at a volume size that gives three volumes (the split runs before any GPU call)."""
    import struct
    from mecat_amd import workload as W
    name, nvols, cells, mcs = W.GRIDS["grid_tiny"]
    n, L, err, G_, seed, ont = W.CONFIGS[name]
    vols = W.synth_volumes(name, nvols, mcs=mcs)
    codes, lens = W.synth_reads(n, L, err, G_, seed, ont)
    fa = str(tmp_path / "r.fa")
    W.write_fasta(fa, codes, lens)
    wrk = str(tmp_path / "wrk")
    subprocess.run([BIN, "-j", "0", "-d", fa, "-o", str(tmp_path / "o"), "-w", wrk], capture_output=True, env=dict(os.environ, MECAT_HIP_MCS=str(mcs)))
    names = open(os.path.join(wrk, "fileindex.txt")).read().split()
    assert len(names) == len(vols) == nvols
    for v, p in zip(vols, names):
        mine = struct.pack("<iii", len(v["lens"]), v["num_bases"], v["start_read_id"]) + v["offs"].tobytes() + v["pac"].tobytes()
        assert mine == open(p, "rb").read(), p
    assert sum(len(v["lens"]) for v in vols) == n


def test_a_rank_that_starts_after_a_peer_died_still_notices(tmp_path):
    """rank 0 dies on a bad input and leaves its failure marker; a rank of the same attempt that starts seconds later (launcher
    skew) must not ignore that marker for being older than itself — rank 0 also removed its heartbeat, so nothing else would end the
    wait for the split marker (6 h by default).  An older marker is believed once it has outlived the grace period."""
    import time
    wrk = tmp_path / "w"
    wrk.mkdir()
    env = dict(os.environ, MECAT_HIP_WORLD="2", MECAT_HIP_RUN_ID="skew", MECAT_HIP_PEER_GRACE_S="2", MECAT_HIP_NO_RESERVE="1")
    args = [BIN, "-j", "0", "-d", str(tmp_path / "missing.fa"), "-o", str(tmp_path / "o"), "-w", str(wrk)]
    r0 = subprocess.run(args, capture_output=True, text=True, env=dict(env, MECAT_HIP_RANK="0"), timeout=60)
    assert r0.returncode != 0
    assert (wrk / "rank_0.failed.xskew").exists()
    time.sleep(2.5)                                # more than the one second the marker's age is compared with
    t0 = time.time()
    r1 = subprocess.run(args, capture_output=True, text=True, env=dict(env, MECAT_HIP_RANK="1"), timeout=60)
    assert r1.returncode != 0 and time.time() - t0 < 30, (r1.returncode, r1.stderr[-300:])
    assert "a peer left a failure marker" in r1.stderr
