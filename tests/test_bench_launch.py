"""`python3 bench.py --gpus N` started WITHOUT a launcher (the way the driver starts the N = 1 run) must produce
the N-rank run by itself: the parent starts torch.distributed.run as a child before it touches a GPU, passes rank
0's line through and leaves with the child's exit code (it used to assert).  CPU test: the
ranks run bench.py's launch self-test (gloo, no GPU); the real two-rank product on one GPU is
tests/test_gpu_multirank.py::test_bench_starts_its_own_ranks."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, rc="0"):
    env = dict(os.environ)
    env.update({"SPX_BENCH_LAUNCH_SELFTEST": "1", "SPX_BENCH_LAUNCH_SELFTEST_RC": rc, "SPX_BENCH_BACKEND": "gloo"})
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2",
                           "--warmup", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_bench_starts_two_ranks_itself():
    p = run(2)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines == [{"selftest": True, "n_gpus": 2, "rank_sum": 3.0}], p.stdout
    assert "without a launcher" in p.stderr


def test_bench_passes_a_failing_rank_on():
    p = run(2, rc="7")
    assert p.returncode != 0


def test_parent_does_not_touch_the_gpu_or_exec():
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def self_launch"):src.index("def main()")]
    assert "import torch" not in body and not re.search(r"os\.exec\w*\(", body) and "subprocess.Popen" in body
    main = src[src.index("def main()"):]
    assert main.index("self_launch(args.gpus") < main.index("import torch")
