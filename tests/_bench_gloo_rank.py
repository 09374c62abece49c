"""Rank entry used by tests/test_bench_selflaunch.py ONLY: what bench.py's self-launch starts per rank, with the gloo process
group and the CPU oracle as evaluator instead of the HIP backend (this container has no GPU).  bench.py itself never
imports this file; its own self-launch always starts bench.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from test_bench_flow_gloo import GlooOracleBackend  # noqa: E402

if __name__ == "__main__":
    if os.environ.get("NMGP_TEST_FAIL_RANK") == os.environ.get("RANK"):
        print("rank %s: failing on purpose" % os.environ.get("RANK"), flush=True)
        sys.exit(3)
    if os.environ.get("NMGP_TEST_HANG_RANK") == os.environ.get("RANK"):
        import time
        print("rank %s: hanging on purpose" % os.environ.get("RANK"), flush=True)
        time.sleep(3600)
    bench.main(sys.argv[1:], backend=GlooOracleBackend())
