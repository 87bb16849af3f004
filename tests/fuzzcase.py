"""Random small search problems for the hypothesis-driven parity tests: tiny
vocabularies and embedding dimensions so that approximate matches, distance
ties, full NearestFilter lists and out-of-vocabulary vectors are common."""

import numpy as np

from fandom_search_amd import abi
from fandom_search_amd.vocab import pack_strings


def make_case(seed, n, H, B, D, V, unique, thr, one_hot, oov_rate, n_script, works):
    rng = np.random.default_rng(seed)
    if one_hot:
        D = max(D, V)
        emb = np.zeros((V, D), dtype=np.float32)
        emb[np.arange(V), rng.permutation(D)[:V]] = 1.0
    else:
        emb = rng.standard_normal((V, D)).astype(np.float32)
        for i in range(1, V, 3):                     # near-synonyms
            emb[i] = emb[i - 1] + 0.15 * rng.standard_normal(D).astype(np.float32)
        if V > 4:
            emb[V - 1] = emb[V - 2]                  # identical vectors, different ids
    normals = rng.standard_normal((H, B, D * n))
    strings = ["w%d" % i for i in range(V)] + ["W%d" % i for i in range(V)]
    n_oov = 6

    def oov_id():
        a, b, c = sorted(int(x) for x in rng.integers(0, D, size=3))
        return abi.FS_OOV_FLAG | ((a * D + b) * D + c)

    oov = [oov_id() for _ in range(n_oov)]
    strings += ["Oov%d" % i for i in range(n_oov)]

    def draw(count):
        vec = rng.integers(0, V, size=count).astype(np.uint32)
        sid = vec + np.uint32(V) * rng.integers(0, 2, size=count).astype(np.uint32)
        for i in np.nonzero(rng.random(count) < oov_rate)[0]:
            k = int(rng.integers(0, n_oov))
            vec[i] = oov[k]
            sid[i] = 2 * V + k
        return vec, sid

    script, script_sid = draw(n_script)
    toks, sids = [], []
    for ln in works:
        v, s = draw(ln)
        for _ in range(2):                            # planted script spans
            if ln >= n and n_script >= n:
                span = int(rng.integers(n, min(ln, n_script, 3 * n) + 1))
                src = int(rng.integers(0, n_script - span + 1))
                dst = int(rng.integers(0, ln - span + 1))
                v[dst:dst + span] = script[src:src + span]
                s[dst:dst + span] = script_sid[src:src + span]
                if rng.random() < 0.5 and not one_hot:
                    k = dst + int(rng.integers(0, span))     # synonym swap: near match
                    if not (v[k] & abi.FS_OOV_FLAG):
                        v[k] = (int(v[k]) ^ 1) % V
        toks.append(v)
        sids.append(s)
    off = np.zeros(len(works) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(works)
    tok = np.concatenate(toks) if toks else np.zeros(0, np.uint32)
    tok_str = np.concatenate(sids) if sids else np.zeros(0, np.uint32)
    cfg = abi.make_config(window_size=n, number_of_hashes=H, hash_dimensions=B,
                          distance_threshold=thr, emb_dim=D, unique_filter=unique)
    chars, coff = pack_strings(strings)
    swords = [strings[int(s)].lower() for s in script_sid]
    return dict(cfg=cfg, emb=emb, normals=normals, script=script, swords=swords,
                tok=tok.astype(np.uint32), tok_str=tok_str.astype(np.uint32), off=off,
                chars=chars, coff=coff, strings=strings)
