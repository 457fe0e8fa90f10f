import sys, os, time
sys.path.insert(0, "/root/repo")
from bench import build_scene
rs = []
for rank in (0, 3, 5, 7):
    r, msg, info = build_scene(1920, 1080, 512, rank, 8, 0)
    r.bind_uniforms(); r.render(frames=4, rebind=False); r.finish()
    rs.append(r)
def run(group, frames=40):
    for r in rs: r.finish()
    t0 = time.perf_counter()
    for f in range(frames):
        for r in group:
            r.render(frames=1, rebind=False)
    for r in group: r.finish()
    return (time.perf_counter() - t0) / frames * 1e3
for n in (1, 2, 4):
    print(f"{n} contexts interleaved on their own streams: {run(rs[:n]):.4f} ms per round ({run(rs[:n])/n:.4f} ms per frame-shard)")
