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
