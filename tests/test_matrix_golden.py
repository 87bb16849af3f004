"""`matrix` against fixtures written by the reference's own StrictNgramDedupe
(/root/reference/_deprecated.py:91-302, run by tests/golden/make_matrix_golden.py in the
build container): the only reference-pinned part of this repo.  The rewrite in
fandom_search_amd/matrix.py must produce the same bytes; with a GPU, so must the whole
chain HIP search -> dated match CSV -> matrix."""

import csv
import glob
import io
import os
import re

import numpy as np
import pytest

from fandom_search_amd import matrix, synth
from tests import util

CASES = sorted(
    (re.match(r"matrix_(.+)\.n(\d+)\.csv$", os.path.basename(p)).groups())
    for p in glob.glob(os.path.join(util.GOLDEN, "matrix_*.n*.csv")))


def _fixture(name, n):
    with open(os.path.join(util.GOLDEN, "matrix_%s.n%s.csv" % (name, n)), newline="") as fh:
        return fh.read()


def test_fixtures_exist():
    assert len(CASES) >= 7 and {"spans_a", "spans_b", "spans_c"} <= {c[0] for c in CASES}


@pytest.mark.parametrize("name,n", CASES)
def test_matrix_matches_the_reference_output(name, n, tmp_path):
    src = os.path.join(util.GOLDEN, "matrix_%s.in.csv" % name)
    out = tmp_path / "m.csv"
    matrix.StrictNgramDedupe(src, ngram_size=n).write_match_work_count_matrix(str(out))
    assert open(out, newline="").read() == _fixture(name, n)


@pytest.mark.skipif(not os.path.exists("/root/reference/_deprecated.py"),
                    reason="the reference is only present in the build container")
def test_fixtures_are_what_the_reference_writes_today(tmp_path):
    """Regenerates every fixture with the reference and compares (build container only)."""
    from tests.golden import make_matrix_golden as mk
    ref = mk.load_reference()
    seen = 0
    for name, n, text in mk.inputs():
        src = tmp_path / ("%s.in.csv" % name)
        src.write_text(text, encoding="utf-8")
        assert open(os.path.join(util.GOLDEN, "matrix_%s.in.csv" % name), newline="").read() == text
        dst = tmp_path / ("%s.out.csv" % name)
        ref.StrictNgramDedupe(str(src), ngram_size=n).write_match_work_count_matrix(str(dst))
        assert open(dst, newline="").read() == _fixture(name, n)
        seen += 1
    assert seen == len(CASES)


def test_no_span_of_n_words(tmp_path):
    """Where the reference dies (IndexError at _deprecated.py:142, `rows[0]` of an empty
    matrix) this implementation writes the two header rows."""
    from tests.golden.make_matrix_golden import span_csv
    src = tmp_path / "in.csv"
    with open(src, "w", newline="") as fh:
        fh.write(span_csv([("a.txt", 0, 10, 3)]))
    assert matrix.StrictNgramDedupe(str(src), 6).matrix_rows() == [["FILENAME"], ["(total)"]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", util.GOLDEN_CASES)
def test_hip_search_then_matrix_matches_the_reference(name, synth_base, tmp_path):
    """HIP search of the golden inputs -> match CSV with header (as `ao3.py search` writes
    it) -> matrix: the bytes the reference's StrictNgramDedupe wrote for the golden records."""
    from fandom_search_amd import search
    from fandom_search_amd.engine import ScriptIndex
    case = util.load_case(name)
    cfg = util.case_config(case)
    normals = synth.lsh_normals(cfg.window_size, cfg.number_of_hashes, cfg.hash_dimensions)
    words = synth_base["words"]
    script = np.asarray(case["script"], dtype=np.uint32)
    ix = ScriptIndex(script, [words[int(t)] for t in script], synth_base["emb"], normals, cfg=cfg)
    tok, off = util.case_arrays(case)
    rows, _ = ix.search(ix.corpus(tok, off, synth_base["chars"], synth_base["off"]))
    ix.close()
    buf = io.StringIO()
    csv.writer(buf).writerow(search.new_record_structure['fields'])
    src = tmp_path / "match.csv"
    with open(src, "w", newline="", encoding="utf-8") as fh:
        fh.write(buf.getvalue() + util.rows_to_csv(rows, case, words))
    out = tmp_path / "m.csv"
    matrix.StrictNgramDedupe(str(src), ngram_size=case["window_size"]) \
        .write_match_work_count_matrix(str(out))
    assert open(out, newline="").read() == _fixture(name, case["window_size"])
