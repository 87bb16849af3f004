"""The plain-C oracle (the parity anchor) under AddressSanitizer and UBSan: a
child process preloads libasan, loads oracle/liboracle_san.so and runs ragged,
empty, OOV and crowded cases; any report fails the test."""

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import ctypes, os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    from oracle import c_oracle
    c_oracle._LIB = None
    real_cdll = ctypes.CDLL
    def cdll(path, *a, **k):
        if path.endswith("liboracle.so"):
            path = path.replace("liboracle.so", "liboracle_san.so")
        return real_cdll(path, *a, **k)
    ctypes.CDLL = cdll
    c_oracle.C.CDLL = cdll
    from fandom_search_amd import abi, synth
    from tests import fuzzcase, util
    words = synth.vocab_words(); emb = synth.embedding()
    from fandom_search_amd.vocab import pack_strings
    chars, coff = pack_strings(words)
    script = synth.script_tokens(900)
    tok, off = util.ragged_corpus([0, 5, 6, 300, 0, 257, 1], script)
    oi = util.oracle_index(abi.make_config(), script, words, emb, synth.lsh_normals(6), threads=3)
    rows, st = oi.search(tok, off, chars, coff)
    assert len(rows) > 0
    oi.close()
    total = 0
    for seed in range(12):
        case = fuzzcase.make_case(seed=seed, n=1 + seed %% 7, H=1 + seed %% 5, B=1 + seed %% 6,
                                  D=[4, 9, 16][seed %% 3], V=3 + seed, unique=bool(seed & 1),
                                  thr=0.3, one_hot=bool(seed & 2), oov_rate=0.1,
                                  n_script=seed * 6, works=[seed, 40, 0])
        sch, so = pack_strings(case["swords"])
        oi = c_oracle.OracleIndex(case["cfg"], case["script"], sch, so, case["emb"],
                                  case["normals"], threads=2)
        r, _ = oi.search(case["tok"], case["off"], case["chars"], case["coff"],
                         tok_str=case["tok_str"])
        total += len(r)
        oi.close()
    print("SANITIZED-OK", len(rows), total)
''')


def test_c_oracle_under_asan_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_san.so"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    child = tmp_path / "child.py"
    child.write_text(CHILD % dict(root=ROOT))
    env = dict(os.environ, LD_PRELOAD=libasan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=67")
    p = subprocess.run([sys.executable, str(child)], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "SANITIZED-OK" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
