"""Host logic of the accept stage (mecat_amd/csrc/aln_strings.h): the table-driven builder of the two gapped strings equals the column-by-column
form on 20 000 random column sets.  CPU only: compiled with g++ on the spot."""
import os
import subprocess

import helpers as H


def test_string_builder_equals_the_column_by_column_form(tmp_path):
    exe = str(tmp_path / "alnstr_check")
    subprocess.run(["g++", "-O2", "-I" + os.path.join(H.ROOT, "mecat_amd", "csrc"), os.path.join(H.ROOT, "tests", "native", "alnstr_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout[-500:]
