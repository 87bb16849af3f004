"""Randomised cross-checks on the GPU (tools/stress_rows.py, tools/stress_lsh.py): a few dozen
random shapes per run; the tools take --cases / --seed for longer sessions."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + list(args),
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches: 0" in out.stdout


def test_scan_rows_against_the_chained_kernels_on_random_shapes():
    _run("stress_rows.py", "--cases", "40", "--seed", "11")


def test_string_id_batches_table_against_per_match_levenshtein():
    _run("stress_rows.py", "--strings", "--cases", "30", "--seed", "12")


def test_lsh_prefilters_against_the_unfiltered_pipeline_on_random_shapes():
    _run("stress_lsh.py", "--cases", "30", "--seed", "13")
