import sys, os
sys.path.insert(0, os.getcwd())
import bench
r, msg, info = bench.build_scene(1920, 1080, 512, 0, 1, 0)
r.bind_uniforms()
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    r.render(frames=32, rebind=False, in_flight=32); r.finish()
for P in (1, 20, 32):
    r.restart_rendering(); r.bind_uniforms()
    r.render(frames=3, rebind=False); r.finish(); r.reset_counters()
    for _ in range(3 if P > 1 else 24):
        r.render(frames=P, rebind=False, in_flight=P)
    r.finish()
    c = r.counters()
    print(f"dvr fpl {P}: {c.kernel_ms / c.frames:.4f} ms/frame util {c.samples / c.lane_slots:.3f} windows/frame {c.gathers / 3 / c.frames:.0f}", flush=True)
