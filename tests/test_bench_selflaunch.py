"""`python bench.py --gpus N` with no torch.distributed environment starts its own ranks (bench.self_launch: a CHILD
`python -m torch.distributed.run`, never an exec of a process that has touched the GPU), relays rank 0's ONE JSON line and
propagates failure.  Here the ranks run tests/_bench_gloo_rank.py (gloo + CPU oracle); the launch code is bench.py's own.
Reference pattern: `mpirun -n 40 python Nonseparable_model_mpisim.py` (sim_job:9), rank -> subject (Nonseparable_model_mpisim.py:41-43,305-306)."""
import io
import json
import os
import subprocess
import sys

from conftest import ROOT

RANK_SCRIPT = os.path.join(ROOT, "tests", "_bench_gloo_rank.py")
ARGS = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--N", "14", "--M", "2", "--chains", "2", "--grad-steps", "0",
        "--no-cpu-baseline"]


def test_self_launch_gives_exactly_one_line_with_world_size_2(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    buf = io.StringIO()
    rc = bench.self_launch(ARGS, 2, script=RANK_SCRIPT, out=buf, timeout=600)
    assert rc == 0
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["chains_total"] == 4 and rec["config"]["chains_ok"] == 4
    dr = rec["distributed"]
    assert dr["process_group"]["world_size"] == 2 and dr["process_group"]["backend"] == "gloo" and dr["self_launched"] is True
    assert [r["rank"] for r in dr["ranks"]] == [0, 1] and [r["local_rank"] for r in dr["ranks"]] == [0, 1]
    assert len({r["pid"] for r in dr["ranks"]}) == 2 and len(dr["ms_per_step_by_rank"]) == 2


def test_self_launch_propagates_a_failing_rank(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("NMGP_TEST_FAIL_RANK", "1")
    buf = io.StringIO()
    rc = bench.self_launch(ARGS, 2, script=RANK_SCRIPT, out=buf, timeout=600)
    assert rc != 0 and buf.getvalue() == ""


def test_self_launch_kills_ranks_that_hang(monkeypatch):
    """A rank that never finishes (first contact with RCCL is the case this guards) keeps the child's stdout open: the timeout must
    still fire, the launcher child must be killed and a non-zero code returned."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("NMGP_TEST_HANG_RANK", "1")
    buf = io.StringIO()
    t0 = time.time()
    rc = bench.self_launch(ARGS, 2, script=RANK_SCRIPT, out=buf, timeout=25)
    assert rc == 124 and buf.getvalue() == "" and time.time() - t0 < 120


def test_bench_main_becomes_the_launcher_before_touching_the_gpu(monkeypatch):
    """bench.py --gpus 2 as a program, without WORLD_SIZE: main() must call self_launch (and nothing else) -- in this container the
    ranks then fail loudly because there is no GPU, and the parent exits non-zero without having imported torch."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--N', '14', '--M', '2'];\n"
            "import bench\n"
            "called = {}\n"
            "def fake(argv, n, **kw):\n"
            "    called['argv'], called['n'] = argv, n\n"
            "    called['torch_loaded'] = 'torch' in sys.modules\n"
            "    return 0\n"
            "bench.self_launch = fake\n"
            "try:\n"
            "    bench.main()\n"
            "except SystemExit as e:\n"
            "    print('EXIT', e.code, called['n'], called['torch_loaded'], called['argv'][:2])\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "EXIT 0 2 False ['--gpus', '2']" in out.stdout
