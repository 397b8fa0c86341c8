"""SURVEY.md §8f row N3 (mecat2canu's mecat2asmpw / mecat2trimpw), first step: the reference pin.  tests/golden/asmpw.json and the
*.S2.sorted files hold what the UNMODIFIED tools (oracle/_ref/mecat2asmpw, mecat2trimpw: gcc on the reference's single C files)
print for a seeded set of corrected reads laid out as canu lays out its overlap blocks.  Where the reference binaries exist
(the build container) this test re-runs them and checks the committed goldens; everywhere it checks the fixtures' integrity.
The restatement and the kernels for this path are not written yet (DESIGN.md §6 lists the deltas to mecat2pw)."""
import hashlib
import json
import os
import sys

import pytest

import helpers as H

META = json.load(open(os.path.join(H.GOLDEN, "asmpw.json")))


def test_committed_outputs_match_their_hashes():
    for name, m in META["outputs"].items():
        p = os.path.join(H.GOLDEN, name)
        if os.path.exists(p):
            txt = open(p).read()
            assert hashlib.sha256(txt.encode()).hexdigest() == m["sha256"] and len(txt.splitlines()) == m["lines"]
    assert META["outputs"]["mecat2asmpw.S1.sorted"]["lines"] > 5000


@pytest.mark.skipif(not os.path.exists(os.path.join(H.ROOT, "oracle", "_ref", "mecat2asmpw")), reason="reference binaries are built in the build container only")
def test_reference_tools_reproduce_the_goldens(tmp_path):
    sys.path.insert(0, H.GOLDEN)
    import make_golden_asmpw as G
    assert G.GEN == META["gen"]
    for tool in ("mecat2asmpw", "mecat2trimpw"):
        d = str(tmp_path / tool)
        os.makedirs(d)
        G.layout(d)
        lines = G.run(tool, d, 2)
        m = META["outputs"]["%s.S2.sorted" % tool]
        assert len(lines) == m["lines"] and hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest() == m["sha256"]
        assert lines == open(os.path.join(H.GOLDEN, "%s.S2.sorted" % tool)).read().splitlines()
