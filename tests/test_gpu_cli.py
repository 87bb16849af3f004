"""End to end through the CLI on the GPU: BASELINE.json configs[0]
(`ao3.py search` on 50 synthetic 1k-token works vs a 500-line script), CSV
bytes compared with a CSV assembled from the C oracle's rows."""

import csv
import datetime
import io
import os

import numpy as np
import pytest

from fandom_search_amd import abi, search, synth, vocab
from fandom_search_amd.cli import main
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("native_csv", ["1", "0"])      # batch files by fs_csvw_* / by csv.writer in forked workers
def test_ao3_search_c1(tmp_path, monkeypatch, synth_base, capsys, native_csv):
    monkeypatch.setenv("FANDOM_SEARCH_NATIVE_CSV", native_csv)
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(5000)
    fandir = tmp_path / "fanworks"
    synth.write_corpus(str(fandir), 50, 1000, script, words)
    (tmp_path / "script.txt").write_text(synth.script_markup(script, words))
    monkeypatch.chdir(tmp_path)
    search.set_vocab(None)
    monkeypatch.delenv("FANDOM_SEARCH_VECTORS", raising=False)
    monkeypatch.delenv("FANDOM_SEARCH_SYNTHETIC_VOCAB", raising=False)
    # no vector table named: refuse instead of silently searching with random vectors
    with pytest.raises(RuntimeError, match="no vector table"):
        main(["search", str(fandir), str(tmp_path / "script.txt")])
    assert main(["search", str(fandir), str(tmp_path / "script.txt"), "--synthetic-vocab"]) == 0
    assert "Processing cluster 0 (0-500)" in capsys.readouterr().out
    today = '{:%Y%m%d}'.format(datetime.date.today())
    final = (tmp_path / ("match-6gram-%s.csv" % today)).read_bytes()
    batch = (tmp_path / "match-6gram-batch-0.csv").read_bytes()
    header = (",".join(search.new_record_structure['fields']) + "\r\n").encode()
    assert final == header + batch and len(batch) > 0

    # expected: the oracle over the same works in the reference's work order
    files = search.list_fan_works(str(fandir))
    order = [int(os.path.basename(f)[1:8]) for f in files]
    tok = np.concatenate([synth.fanwork_tokens(i, 1000, script) for i in order])
    off = np.arange(51, dtype=np.uint64) * np.uint64(1000)
    cfg = abi.make_config()
    oi = util.oracle_index(cfg, script, words, emb, synth.lsh_normals(6))
    rows, _ = oi.search(tok, off, synth_base["chars"], synth_base["off"])
    scene, char = synth.script_columns(len(script))
    buf = io.StringIO()
    wr = csv.writer(buf)
    for r in rows:
        w, f, o = int(r["work"]), int(r["fan_ix"]), int(r["orig_ix"])
        fw, ow = words[tok[w * 1000 + f]], words[script[o]]
        wr.writerow([files[w], f, fw, vocab.hash_string(fw), o, ow, vocab.hash_string(ow),
                     char[o], int(scene[o]), float(r["dist"]), int(r["lev"]), float(r["comb"])])
    assert batch == buf.getvalue().encode()

    # a second run the same day must not clobber the first result
    assert main(["search", str(fandir), str(tmp_path / "script.txt"), "-n", "5", "-s", "3"]) == 0
    second = (tmp_path / ("match-6gram-%s-1.csv" % today)).read_bytes()
    assert second.startswith(header) and second != final

    # matrix over the result
    assert main(["matrix", "match-6gram-%s.csv" % today, "synth"]) == 0
    with open("synth-most-common-perfect-matches-no-overlap-6-gram-match-matrix.csv") as fh:
        m = list(csv.reader(fh))
    assert m[0][0] == "FILENAME" and m[1][0] == "(total)" and len(m[0]) > 1
