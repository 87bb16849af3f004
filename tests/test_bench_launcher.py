"""bench.py --gpus N typed without a launcher starts its own ranks: the
decision is a pure function (tested here without a GPU); with a GPU the command itself
runs two ranks over gloo on the one card and must print one JSON line."""

import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_decision():
    b = _bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = b.launcher_command(b.parse(argv), argv, {}, port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == argv
    # a rank under a launcher, and N = 1, run in this process
    assert b.launcher_command(b.parse(argv), argv, {"WORLD_SIZE": "8"}) is None
    assert b.launcher_command(b.parse(["--gpus", "1"]), ["--gpus", "1"], {}) is None
    assert b.launcher_command(b.parse([]), [], {}) is None
    # an image that merely exports WORLD_SIZE=1 has not launched anything
    assert b.launcher_command(b.parse(argv), argv, {"WORLD_SIZE": "1"}) is not None
    assert b.launcher_command(b.parse(argv), argv, {"WORLD_SIZE": "1", "RANK": "0"}) is None
    # no port given: the launcher's own rendezvous finds a free one
    cmd = b.launcher_command(b.parse(argv), argv, {})
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert "--master-port" not in cmd


def test_defaults_follow_the_contract():
    b = _bench()
    a = b.parse([])
    assert a.gpus == 1 and a.scaling == ""      # N = 1: c2; N > 1: configs[2] split N ways (strong)


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("gather_root,scaling", [("rotate", ""), ("step", ""), ("0", ""), ("rotate", "weak")])
def test_bench_starts_its_own_ranks(tmp_path, gather_root, scaling):
    """`python bench.py --gpus 2 ...` exactly as the driver types it (no launcher): two ranks
    (gloo between them, both on this one GPU), one JSON line from rank 0, the records every
    rank sent checked on the rank that received the last step -- for the all_to_all per N
    steps (default), a gather per step with the root going round, and a gather to rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--works", "200", "--steps", "3", "--warmup", "1", "--reps", "2",
                        "--gather-root", gather_root] + (["--scaling", scaling] if scaling else []),
                       env=env, cwd=str(tmp_path), timeout=800, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    # typed as is, N > 1 measures configs[2] (100k works x 5k tokens) split over the ranks
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == (scaling or "strong")
    assert d["config"]["tokens_per_work"] == (2000 if scaling == "weak" else 5000)
    assert d["config"]["gather_verified"] is True
    assert d["value"] > 0 and len(d["samples_ms"]) == 2
