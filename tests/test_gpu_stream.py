"""Streamed corpora (BASELINE configs[4]): batches uploaded on a copy stream
while the previous batch is searched give the same rows as one monolithic
search; device-side validation of uploaded ids."""

import numpy as np
import pytest

from fandom_search_amd import _lib, abi, synth
from tests import util

pytestmark = pytest.mark.gpu


def test_streamed_batches_equal_monolithic(synth_base):
    from fandom_search_amd.engine import PinnedBuffer, ScriptIndex, search_stream
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(5000)
    n_works, tpw, per_batch = 2400, 1000, 400
    tok, off = synth.corpus_tokens(n_works, tpw, script)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    whole, st = ix.search(ix.corpus(tok, off, synth_base["chars"], synth_base["off"]))
    # batches of different sizes, staged in pinned memory (two staging buffers)
    sizes = [per_batch, per_batch // 2, per_batch * 2, per_batch, per_batch + 37,
             n_works - (5 * per_batch + per_batch // 2 + 37)]
    assert sum(sizes) == n_works
    stage_tok = [PinnedBuffer(max(sizes) * tpw, np.uint32) for _ in range(2)]
    stage_off = [PinnedBuffer(max(sizes) + 1, np.uint64) for _ in range(2)]

    def batches():
        w0 = 0
        for i, nb in enumerate(sizes):
            t = stage_tok[i & 1].array[:nb * tpw]
            o = stage_off[i & 1].array[:nb + 1]
            t[:] = tok[w0 * tpw:(w0 + nb) * tpw]
            o[:] = off[w0:w0 + nb + 1] - off[w0]
            yield t, o
            w0 += nb

    parts, w0, windows = [], 0, 0
    for (rows, bst), nb in zip(search_stream(ix, batches(), synth_base["chars"],
                                             synth_base["off"]), sizes):
        rows = rows.copy()
        rows["work"] += np.uint32(w0)
        parts.append(rows)
        windows += bst.windows_processed
        w0 += nb
    got = np.concatenate(parts)
    assert got.tobytes() == whole.tobytes()
    assert windows == st.windows_processed
    # and against the oracle on the first batch
    oi = util.oracle_index(abi.make_config(), script, words, emb, synth.lsh_normals(6))
    want, _ = oi.search(tok[:sizes[0] * tpw], off[:sizes[0] + 1], synth_base["chars"],
                        synth_base["off"])
    util.assert_rows_equal(parts[0], want)


def test_device_side_validation(synth_base):
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(500)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    tok = np.arange(100, dtype=np.uint32)
    off = np.array([0, 100], dtype=np.uint64)
    bad = tok.copy()
    bad[57] = 8192                                   # one past the vector table
    with pytest.raises(_lib.FsError, match="outside the vector table"):
        ix.corpus(bad, off, synth_base["chars"], synth_base["off"])
    short_chars, short_off = synth_base["chars"][:40], synth_base["off"][:11]   # 10 strings
    with pytest.raises(_lib.FsError, match="outside the string table"):
        ix.corpus(tok, off, short_chars, short_off)
    oov = tok.copy()
    oov[3] = abi.FS_OOV_FLAG | 5
    with pytest.raises(_lib.FsError, match="need string ids"):
        ix.corpus(oov, off, synth_base["chars"], synth_base["off"])
    with pytest.raises(_lib.FsError, match="work_off"):
        ix.corpus(tok, np.array([0, 60, 50], dtype=np.uint64), synth_base["chars"],
                  synth_base["off"])
    # a good corpus still works afterwards
    rows, _ = ix.search(ix.corpus(tok, off, synth_base["chars"], synth_base["off"]))
    assert len(rows) == 0


def test_update_of_a_corpus_with_a_search_in_flight_is_refused(synth_base):
    """fs_corpus_update_begin while a search of the same corpus has been queued and not
    finished would race with its kernels: FS_E_INVALID until fs_search_corpus_end."""
    import torch
    from fandom_search_amd import _lib
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(2000)
    tok, off = util.ragged_corpus([600] * 10, script)
    tok2, off2 = util.ragged_corpus([500] * 8, script, first_work=50)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    c = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
    want, _ = ix.search(c)
    buf = torch.zeros(32 + (len(want) + 8) * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # torch's fill / copy is complete before the library's own streams touch the buffer (engine.torch_ready)
    t = ix.search_begin(c, buf.data_ptr(), len(want) + 8, header=True)
    with pytest.raises(_lib.FsError) as e:
        c.update_begin(tok2, off2)
    assert e.value.code == abi.FS_E_INVALID
    n, _ = ix.search_end(t)
    assert n == len(want)
    assert buf[32:32 + n * 32].cpu().numpy().tobytes() == want.tobytes()
    c.update_begin(tok2, off2)            # fine now
    c.update_end()
    got2, _ = ix.search(c)
    oi = util.oracle_index(abi.make_config(), script, words, emb, synth.lsh_normals(6))
    want2, _ = oi.search(tok2, off2, synth_base["chars"], synth_base["off"])
    util.assert_rows_equal(got2, want2)
    ix.close()


@pytest.mark.timeout(1500)
def test_configs4_at_full_size(synth_base):
    """BASELINE.json configs[4] in full: 1 000 000 works x 1 000 tokens (4 GB of ids) streamed
    from pinned host memory in ten batches through two device corpora.  Too large for the
    oracle: the records are checked through size-independent properties (one record per
    word, ascending, the matched script word carries the fan word's vector id, the window
    count), two of the ten batches against the numpy n-gram join, and the first batch's
    first 300 works against the oracle."""
    from fandom_search_amd.engine import PinnedBuffer, ScriptIndex, search_stream
    from tests.test_gpu_fullsize import _check_rows, _covered_words
    conf = synth.CONFIGS["c5"]
    n_works, tpw, n_batches, n = conf["n_works"], conf["tokens_per_work"], 10, 6
    per = n_works // n_batches
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(conf["script_tokens"])
    normals = synth.lsh_normals(n)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, normals, cfg=abi.make_config())
    stage_tok = [PinnedBuffer(per * tpw, np.uint32) for _ in range(2)]
    stage_off = [PinnedBuffer(per + 1, np.uint64) for _ in range(2)]
    off = np.arange(per + 1, dtype=np.uint64) * np.uint64(tpw)
    kept = {}

    def batches():
        for b in range(n_batches):
            tok, _ = synth.corpus_tokens_parallel(per, tpw, script, first_work=b * per)
            if b in (0, 7):
                kept[b] = tok.copy()
            stage_tok[b & 1].array[:] = tok
            stage_off[b & 1].array[:] = off
            yield stage_tok[b & 1].array, stage_off[b & 1].array

    total_rows, windows = 0, 0
    for b, (rows, st) in enumerate(search_stream(ix, batches(), synth_base["chars"], synth_base["off"])):
        assert st.path == abi.FS_MODE_EXACT and st.scan_launches == 1
        assert st.windows_processed == per * (tpw - n + 1)
        windows += st.windows_processed
        total_rows += len(rows)
        pos = off[rows["work"]].astype(np.int64) + rows["fan_ix"].astype(np.int64)
        assert np.all(np.diff(pos) > 0)                       # one record per word, ascending
        assert np.all(rows["lev"] == n + 1) and np.all(np.abs(rows["dist"]) < 1e-15)
        if b in kept:
            got_pos = _check_rows(rows, kept[b], off, script, n)
            want_pos, _ = _covered_words(kept[b], off, script, n)
            assert np.array_equal(got_pos, want_pos)          # completeness, no extras
        if b == 0:
            oi = util.oracle_index(abi.make_config(), script, words, emb, normals)
            want, _ = oi.search(kept[0][:300 * tpw], off[:301], synth_base["chars"], synth_base["off"])
            util.assert_rows_equal(rows[rows["work"] < 300], want)
    assert windows == n_works * (tpw - n + 1) == 995_000_000
    assert total_rows > 10_000_000
    for pb in stage_tok + stage_off:
        pb.close()
