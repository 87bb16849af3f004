"""Synthetic inputs for the 6-gram search path.

The reference ships no script, no fan works and no vectors (its data dirs hold
only .gitignore files), and its LSH hyperplanes are unseeded
(/root/reference/search.py:114-115), so every input the search consumes is
generated here from fixed seeds and handed to both the oracle and the HIP
library:

* vocabulary   8192 lower-case alphabetic pseudo-words (one token under any
               tokenizer), word id == vector row id
* embedding    E[V][300] float32, unit-normalised N(0,1),
               numpy.random.default_rng(20240601)
* normals      default_rng(4815162342).standard_normal((H, B, 300*n)) float64
               (the seed the reference uses for its work shuffle,
               search.py:354)
* script       i.i.d. Zipf(s=1.1) draws over V, 10 tokens per LINE<<>>,
               CHARACTER_NAME cycling over 8 names, SCENE_NUMBER every 50 lines
* fan works    Zipf draws with planted reuse: per work Poisson(2) verbatim
               script spans of length U[6,24]; 10 % of plants get one token
               substituted (a near miss that must NOT match at that token);
               per-work seed 1_000_003 * work_idx + 17; names w%07d.txt
"""

import os

import numpy as np

VOCAB_SIZE = 8192
EMB_DIM = 300
EMB_SEED = 20240601
NORMALS_SEED = 4815162342
SCRIPT_SEED = 20240602
ZIPF_S = 1.1
TOKENS_PER_LINE = 10
LINES_PER_SCENE = 50
CHARACTERS = ("ALDER", "BRISA", "CORMAC", "DELPHINE",
              "EAMON", "FREYA", "GARETH", "HESTER")

_NOT_ONE_TOKEN = ("weve",)
_CONS = "bcdfghjklmnprstvwxyz"   # 20
_VOWS = "aeiou"                  # 5  -> 100 syllables, 10000 two-syllable words


def vocab_words(size=VOCAB_SIZE):
    """`size` distinct lower-case alphabetic pseudo-words (CVCV)."""
    syl = [c + v for c in _CONS for v in _VOWS]
    if size > len(syl) ** 2:
        raise ValueError("vocabulary too large for two syllables")
    words = [syl[i // len(syl)] + syl[i % len(syl)] for i in range(size)]
    # "weve" is a tokenizer exception of spaCy's English rules (we + ve): one token under
    # any tokenizer means it has to go; its slot takes a pseudo-word beyond the 8192
    for i, w in enumerate(words):
        if w in _NOT_ONE_TOKEN:
            words[i] = syl[-1] + syl[-1 - _NOT_ONE_TOKEN.index(w)]
    return words


def embedding(size=VOCAB_SIZE, dim=EMB_DIM, seed=EMB_SEED):
    rng = np.random.default_rng(seed)
    emb = rng.standard_normal((size, dim)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    return np.ascontiguousarray(emb, dtype=np.float32)


def clustered_table(seed=3, clusters=1024, per=8, dim=EMB_DIM, noise=0.25):
    """A table shaped like a real word-embedding table as far as this search is
    concerned: `clusters` groups of `per` near-synonyms (cosine 0.85-0.97 inside a
    group), so approximate matches exist, c_max ~ 1 and the exact-n-gram proof fails
    (the LSH pipeline runs).  Returns (emb[clusters*per][dim] float32, perm): row
    perm[i] holds member i % per of group i // per (synonyms are not neighbours in id
    space)."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((clusters, dim))
    emb = np.repeat(centers, per, axis=0) + noise * rng.standard_normal((clusters * per, dim))
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    perm = rng.permutation(len(emb))
    out = np.empty_like(emb)
    out[perm] = emb
    return np.ascontiguousarray(out, dtype=np.float32), perm


def synonym_swaps(tok, perm, per=8, rate=0.1, seed=9):
    """Replace `rate` of the tokens by a random member of their group of
    `clustered_table` (genuine approximate matches)."""
    rng = np.random.default_rng(seed)
    inv = np.argsort(perm)
    tok = np.array(tok, dtype=np.uint32, copy=True)
    sel = np.nonzero(rng.random(len(tok)) < rate)[0]
    orig = inv[tok[sel]]
    syn = (orig // per) * per + rng.integers(0, per, size=len(sel))
    tok[sel] = perm[syn].astype(np.uint32)
    return tok


def realistic_words(size):
    """`size` distinct lower-case pseudo-words, one token each: vocab_words() and, beyond its
    reach, three-syllable ones."""
    syl = [c + v for c in _CONS for v in _VOWS]
    words = vocab_words(min(size, len(syl) ** 2 - 64))
    seen = set(words)
    i = 0
    while len(words) < size:
        w = syl[i % 100] + syl[(i // 100) % 100] + syl[(i // 10000) % 100]
        i += 1
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


def realistic_table(rows=20000, dim=EMB_DIM, seed=11, sigma=0.3, dup_share=0.05, zero_rows=16):
    """A table with the features of a real word-embedding table (en_core_web_md keeps 20k
    unique rows) that the benchmark tables lack, as far as they matter to this search:
      * vectors are NOT unit length: norms lognormal(log 6, sigma) -- a factor of ~3 between
        the short and the long ones at sigma = 0.3;
      * similarity at several scales: a three-level hierarchy, cosine ~0.95 inside the
        innermost groups (5 words), ~0.8 inside the middle ones (20), ~0.6 inside the outer
        ones (80), ~0 across;
      * `dup_share` of the rows are exact copies of other rows (spaCy prunes vectors: many
        keys share a row's content), and `zero_rows` rows are zero.
    Returns (emb float32 [rows][dim], group): group[r] = (outer, middle, inner) of row r."""
    rng = np.random.default_rng(seed)
    inner, mid, outer = 5, 4, 4                     # words per inner group, inner per middle, middle per outer
    n_inner = -(-rows // inner)
    n_mid = -(-n_inner // mid)
    n_out = -(-n_mid // outer)
    unit = lambda m: (lambda a: a / np.linalg.norm(a, axis=1, keepdims=True))(rng.standard_normal((m, dim)))
    u1, u2, u3 = unit(n_out), unit(n_mid), unit(n_inner)
    r = np.arange(rows)
    gi, gm, go = r // inner, r // (inner * mid), r // (inner * mid * outer)
    a, b, c, d = np.sqrt(0.6), np.sqrt(0.2), np.sqrt(0.15), np.sqrt(0.05)
    emb = a * u1[go] + b * u2[gm] + c * u3[gi] + d * unit(rows)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    emb *= rng.lognormal(np.log(6.0), sigma, size=(rows, 1))
    n_dup = int(rows * dup_share)
    dst = rng.choice(rows, size=n_dup, replace=False)
    src = rng.integers(0, rows, size=n_dup)
    emb[dst] = emb[src]
    emb[rng.choice(rows, size=zero_rows, replace=False)] = 0.0
    perm = rng.permutation(rows)                    # related words are not neighbours in id space
    out = np.empty_like(emb)
    out[perm] = emb
    group = np.empty((rows, 3), dtype=np.int32)
    group[perm] = np.stack([go, gm, gi], axis=1)
    return np.ascontiguousarray(out, dtype=np.float32), group


def realistic_corpus(n_works, tokens_per_work, script, group, n_table, oov_rate=0.08, cap_rate=0.06,
                     swap_rate=0.12, n_oov_words=400, seed=5, first_work=0):
    """Fan works over a `realistic_table`: Zipf text with planted script spans (fanwork_tokens),
    `swap_rate` of the tokens replaced by a word of the same inner group (near-synonym,
    cosine ~0.95) or, one time in four, of the same middle group (~0.8); `oov_rate` of the
    tokens out-of-vocabulary words (names: `n_oov_words` distinct strings without a row -- the
    reference gives them 3-hot vectors, search.py:79-83); `cap_rate` of the in-vocabulary
    tokens capitalised (the fan side is case-sensitive, search.py:166: same row, other
    string).  Returns (tok_str, work_off, strings): string ids into `strings` =
    realistic_words(n_table) + capitalised forms + the OOV words; the caller maps strings to
    vector ids (vocab.Vocab.encode, or vector_ids_of below)."""
    rng = np.random.default_rng(seed + 7919 * first_work)
    tok = np.empty(n_works * tokens_per_work, dtype=np.uint32)
    for i in range(n_works):
        tok[i * tokens_per_work:(i + 1) * tokens_per_work] = fanwork_tokens(
            first_work + i, tokens_per_work, script, n_table)
    # members of every inner / middle group
    order_i = np.argsort(group[:, 2], kind="stable")
    start_i = np.searchsorted(group[order_i, 2], np.arange(group[:, 2].max() + 2))
    order_m = np.argsort(group[:, 1], kind="stable")
    start_m = np.searchsorted(group[order_m, 1], np.arange(group[:, 1].max() + 2))
    sel = np.nonzero(rng.random(len(tok)) < swap_rate)[0]
    wide = rng.random(len(sel)) < 0.25
    for idxs, order, start, col in ((sel[~wide], order_i, start_i, 2), (sel[wide], order_m, start_m, 1)):
        gsel = group[tok[idxs], col]
        lo, hi = start[gsel], start[gsel + 1]
        tok[idxs] = order[lo + (rng.random(len(idxs)) * (hi - lo)).astype(np.int64)].astype(np.uint32)
    tok_str = tok.copy()
    cap = rng.random(len(tok)) < cap_rate
    tok_str[cap] += np.uint32(n_table)                         # capitalised form of the same row
    oov = rng.random(len(tok)) < oov_rate
    tok_str[oov] = (2 * n_table + rng.integers(0, n_oov_words, size=int(oov.sum()))).astype(np.uint32)
    off = np.arange(n_works + 1, dtype=np.uint64) * np.uint64(tokens_per_work)
    return tok_str, off


def realistic_strings(n_table, n_oov_words=400):
    """String table of realistic_corpus: the table's words, their capitalised forms, OOV names."""
    words = realistic_words(n_table)
    syl = [c + v for c in _CONS for v in _VOWS]
    names = ["Q" + syl[i % 100] + "x" + syl[(i // 100) % 100] + "q" for i in range(n_oov_words)]
    return words + [w.capitalize() for w in words] + names


def realistic_vector_ids(n_table, n_oov_words=400, dim=EMB_DIM):
    """Vector id per string of realistic_strings: the row for a table word and its capitalised
    form, the out-of-vocabulary code (3-hot positions by the seeded hash, vocab.Vocab) for a name."""
    from . import vocab as vocab_mod
    strings = realistic_strings(n_table, n_oov_words)
    vid = np.empty(len(strings), dtype=np.uint32)
    vid[:n_table] = np.arange(n_table, dtype=np.uint32)
    vid[n_table:2 * n_table] = np.arange(n_table, dtype=np.uint32)
    h = vocab_mod.default_oov_hash
    for i, w in enumerate(strings[2 * n_table:]):
        hot = sorted((h(w) % dim, h(w * 2) % dim, h(w * 3) % dim))
        vid[2 * n_table + i] = vocab_mod.OOV_FLAG | ((hot[0] * dim + hot[1]) * dim + hot[2])
    return strings, vid


def lsh_normals(window_size=6, number_of_hashes=15, hash_dimensions=14,
                dim=EMB_DIM, seed=NORMALS_SEED):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(
        (number_of_hashes, hash_dimensions, dim * window_size))


def _zipf_cdf(size=VOCAB_SIZE, s=ZIPF_S):
    p = np.arange(1, size + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(p / p.sum())
    cdf[-1] = 1.0
    return cdf


_CDF_CACHE = {}


def _draw(rng, count, size):
    cdf = _CDF_CACHE.get(size)
    if cdf is None:
        cdf = _CDF_CACHE[size] = _zipf_cdf(size)
    ids = np.searchsorted(cdf, rng.random(count), side="right")
    return np.minimum(ids, size - 1).astype(np.uint32)


def script_tokens(n_tokens, vocab_size=VOCAB_SIZE, seed=SCRIPT_SEED):
    return _draw(np.random.default_rng(seed), n_tokens, vocab_size)


def fanwork_tokens(work_idx, n_tokens, script, vocab_size=VOCAB_SIZE):
    """Token ids of synthetic fan work `work_idx` (deterministic)."""
    rng = np.random.default_rng(1_000_003 * int(work_idx) + 17)
    tok = _draw(rng, n_tokens, vocab_size)
    for _ in range(int(rng.poisson(2.0))):
        length = int(rng.integers(6, 25))
        if length > n_tokens or length > len(script):
            continue
        src = int(rng.integers(0, len(script) - length + 1))
        dst = int(rng.integers(0, n_tokens - length + 1))
        span = script[src:src + length].copy()
        if rng.random() < 0.1:
            at = int(rng.integers(0, length))
            span[at] = (int(span[at]) + 1 + int(rng.integers(0, vocab_size - 1))) \
                % vocab_size
        tok[dst:dst + length] = span
    return tok


def corpus_tokens(n_works, tokens_per_work, script, first_work=0,
                  vocab_size=VOCAB_SIZE):
    """Packed token-id buffer + work offsets for works
    [first_work, first_work + n_works)."""
    tok = np.empty(n_works * tokens_per_work, dtype=np.uint32)
    for i in range(n_works):
        tok[i * tokens_per_work:(i + 1) * tokens_per_work] = fanwork_tokens(
            first_work + i, tokens_per_work, script, vocab_size)
    off = np.arange(n_works + 1, dtype=np.uint64) * np.uint64(tokens_per_work)
    return tok, off


def work_name(work_idx):
    return "w%07d.txt" % work_idx


def script_markup(tokens, words):
    """Script in the reference's markup (search.py:290-329): SCENE_NUMBER<<>>,
    CHARACTER_NAME<<>>, LINE<<>> tags, one per text line."""
    out = []
    n_lines = (len(tokens) + TOKENS_PER_LINE - 1) // TOKENS_PER_LINE
    for ln in range(n_lines):
        if ln % LINES_PER_SCENE == 0:
            out.append("SCENE_NUMBER<<%d>>" % (ln // LINES_PER_SCENE + 1))
        out.append("CHARACTER_NAME<<%s>>" % CHARACTERS[ln % len(CHARACTERS)])
        line = tokens[ln * TOKENS_PER_LINE:(ln + 1) * TOKENS_PER_LINE]
        out.append("LINE<<%s>>" % " ".join(words[int(t)] for t in line))
    return "\n".join(out) + "\n"


def script_columns(n_tokens):
    """(scene, character) per script token, as load_markup_script would assign
    them for `script_markup` output."""
    line = np.arange(n_tokens) // TOKENS_PER_LINE
    scene = (line // LINES_PER_SCENE + 1).astype(np.int64)
    character = [CHARACTERS[int(l) % len(CHARACTERS)] for l in line]
    return scene, character


def write_corpus(directory, n_works, tokens_per_work, script, words,
                 first_work=0):
    """Write works as plain-text files (one space between tokens)."""
    os.makedirs(directory, exist_ok=True)
    names = []
    for i in range(first_work, first_work + n_works):
        tok = fanwork_tokens(i, tokens_per_work, script, len(words))
        path = os.path.join(directory, work_name(i))
        with open(path, "w", encoding="utf8") as fh:
            fh.write(" ".join(words[int(t)] for t in tok))
        names.append(path)
    return names


# BASELINE.json configs (script taken as 10 tokens/line).
CONFIGS = {
    "c1": dict(n_works=50, tokens_per_work=1000, script_tokens=5000),
    "c2": dict(n_works=10_000, tokens_per_work=2000, script_tokens=20_000),
    "c3": dict(n_works=100_000, tokens_per_work=5000, script_tokens=20_000),
    "c3shard": dict(n_works=12_500, tokens_per_work=5000, script_tokens=20_000),
    "c5": dict(n_works=1_000_000, tokens_per_work=1000, script_tokens=20_000),
}


def corpus_tokens_parallel(n_works, tokens_per_work, script, first_work=0,
                           vocab_size=VOCAB_SIZE, procs=0):
    """corpus_tokens() over `procs` child processes (same bytes: every work has its own
    seed).  The children are fresh interpreters (`python -m fandom_search_amd.synth`:
    numpy and this module only), so this may be called from a process that has touched
    the GPU, from any kind of __main__."""
    import subprocess
    import sys
    import tempfile
    if not procs:
        # (the ranks of one node share its cores)
        ranks = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1)
        procs = min(16, max(1, len(os.sched_getaffinity(0)) // max(1, ranks)))
    if procs <= 1 or n_works < 4096:
        return corpus_tokens(n_works, tokens_per_work, script, first_work, vocab_size)
    per = -(-n_works // procs)
    tok = np.empty(n_works * tokens_per_work, dtype=np.uint32)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        spath = os.path.join(tmp, "script.npy")
        np.save(spath, np.ascontiguousarray(script, dtype=np.uint32))
        jobs = []
        for lo in range(0, n_works, per):
            n = min(per, n_works - lo)
            out = os.path.join(tmp, "part%d.bin" % lo)
            cmd = [sys.executable, "-m", "fandom_search_amd.synth", spath, out, str(n),
                   str(tokens_per_work), str(first_work + lo), str(vocab_size)]
            jobs.append((lo, n, out, subprocess.Popen(cmd, cwd=root)))
        try:
            for lo, n, out, p in jobs:
                if p.wait() != 0:
                    raise RuntimeError("corpus worker failed (exit code %d)" % p.returncode)
                tok[lo * tokens_per_work:(lo + n) * tokens_per_work] = np.fromfile(out, dtype=np.uint32)
        finally:
            for _, _, _, p in jobs:             # (a failure leaves no child behind)
                if p.poll() is None:
                    p.kill()
                    p.wait()
    off = np.arange(n_works + 1, dtype=np.uint64) * np.uint64(tokens_per_work)
    return tok, off


if __name__ == "__main__":          # worker of corpus_tokens_parallel
    import sys
    _script = np.load(sys.argv[1])
    _n, _tpw, _first, _vs = (int(x) for x in sys.argv[3:7])
    corpus_tokens(_n, _tpw, _script, _first, _vs)[0].tofile(sys.argv[2])
