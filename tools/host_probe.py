import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_scene
r, msg, info = build_scene(256, 256, 64, 0, 1, 0)
r.settings.dvr_max_steps = 1
r.bind_uniforms(); r.render(frames=4, rebind=False); r.finish()
for label, fn in (("python render()", lambda: r.render(frames=1, rebind=False)),
                  ("raw vx_render_frame", lambda: r._lib.vx_render_frame(r._ctx, 7, 0.5))):
    r.finish(); t0 = time.perf_counter()
    for i in range(2000): fn()
    t1 = time.perf_counter(); r.finish(); t2 = time.perf_counter()
    print(f"{label}: {(t1-t0)/2000*1e6:.1f} us per call issued, {(t2-t0)/2000*1e6:.1f} us per frame complete", file=sys.stderr)
c = r.counters(); print("kernel ms avg", c.kernel_ms / c.launches, file=sys.stderr)
