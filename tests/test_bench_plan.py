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
