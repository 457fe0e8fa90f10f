"""bench.py's launch plan (CPU): how K steps of 32 accumulation frames become launches and exchange steps on 1, 2, 4 and 8
ranks -- the part of the multi-GPU bench that can be checked without GPUs."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_driver_command_plan(world):
    b = _bench()
    per_launch = min(64, b.DEFAULT_FRAMES_PER_LAUNCH * world)
    warm, timed = 5 * 32, 20 * 32                       # --warmup 5 --steps 20, 32 frames per step
    plan_w = b.launch_plan(0, warm, per_launch, 64)
    plan_t = b.launch_plan(warm, timed, per_launch, 64)
    for plan, first, count in ((plan_w, 0, warm), (plan_t, warm, timed)):
        assert plan[0][0] == first and sum(n for _, n, _ in plan) == count
        assert all(0 < n <= per_launch for _, n, _ in plan)
        assert max(n for _, n, _ in plan) - min(n for _, n, _ in plan) <= 1          # equal launches
        for (f0, n, _), (g0, _, _) in zip(plan, plan[1:]):
            assert g0 == f0 + n                                                     # consecutive frames, none twice
    assert len(plan_t) == 640 // per_launch
    # the exchange runs once per 64 frames: ten times in the timed region; with two or more ranks (64-frame launches)
    # behind every launch, the last one included (with one rank bench.py adds the exchange of the last 32 frames)
    assert sum(1 for _, _, x in plan_t if x) == 10
    if world > 1:
        assert all(x for _, _, x in plan_t)
    assert any(x for _, _, x in plan_w)                 # the warm-up performs one too (RCCL's first-use set-up)


def test_plan_edges():
    b = _bench()
    assert b.launch_plan(0, 20, 16, 64) == [(0, 10, False), (10, 10, False)]
    assert b.launch_plan(60, 8, 64, 64) == [(60, 8, True)]
    assert b.launch_plan(0, 0, 32, 64) == []
    p = b.launch_plan(3, 1000, 64, 64)
    assert sum(n for _, n, _ in p) == 1000 and all(n <= 64 for _, n, _ in p)


def test_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks as child processes (the parent never touches
    the GPU).  Here there is no GPU: the ranks must reach the no-GPU exit with RANK / WORLD_SIZE set, the whole command
    must fail, and nothing may be printed on stdout -- never a 1-rank result reported for a 2-GPU command."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "launching 2 ranks" in p.stderr
    ranks = {int(m) for m in __import__("re").findall(r"needs an MI355X.*\(rank (\d) of 2\)", p.stderr)}
    assert ranks and ranks <= {0, 1}          # torchrun may end the second rank as soon as the first one fails


def test_world_size_must_match_gpus():
    """a launcher that sets WORLD_SIZE to something other than --gpus is refused, not silently accepted"""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and p.stdout.strip() == ""
