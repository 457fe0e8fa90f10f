"""Generates the committed golden fixtures from the CPU oracle.

The reference has no tests, golden images or vectors and cannot run in
the build image, so these fixtures are outputs of oracle/vx_oracle.c (PARITY UNPINNED).
They pin the oracle against regressions and give the GPU parity tests fixed targets.

    python -m tests.golden.make_golden      # rewrites tests/golden/*.npz, rng_kat.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name -> (volume kind, n, image (w,h), mode, extra settings)
CASES = {
    "sphere32_debughits": ("sphere", 32, (40, 32), "dvr", dict(debug_hits=True)),
    "sphere32_dvr": ("sphere", 32, (40, 32), "dvr", dict()),
    "noise32_dvr_clip": ("noise", 32, (48, 40), "dvr",
                         dict(clip_min=(0.25, 0, 0), clip_max=(1, 1, 0.75), bench=True)),
    "noise32_dvr_jitter_f3": ("noise", 32, (32, 32), "dvr", dict(dvr_jitter=True, frame=3, bench=True)),
    "noise32_phong": ("noise", 32, (32, 32), "dvr_phong", dict(bench=True)),
    "noise32_phong_jitter_f2": ("noise", 32, (40, 32), "dvr_phong", dict(bench=True, dvr_jitter=True, frame=2,
                                                                         clip_min=(0.25, 0, 0), clip_max=(1, 1, 0.75))),
    # [build] orthographic camera (BASELINE config 1)
    "sphere32_dvr_ortho": ("sphere", 32, (40, 32), "dvr", dict(ortho=0.6)),
    "noise32_dvr_ortho_jitter_f1": ("noise", 32, (40, 32), "dvr", dict(bench=True, ortho=0.7, dvr_jitter=True, frame=1)),
    "noise32_raymarch": ("noise", 32, (32, 32), "raymarch", dict(bench=True, frame=2)),
    "noise32_no_dda": ("noise", 32, (32, 32), "no_dda", dict(bench=True, frame=1)),
    "noise32_default": ("noise", 32, (32, 32), "default", dict(bench=True, frame=4)),
    "noise32_default_b3": ("noise", 32, (24, 24), "default", dict(bench=True, frame=6, bounces=3)),
    # with the viewer's default environment map resident (use_env = 1, environment.ts:102-130)
    "sphere32_debughits_env": ("sphere", 32, (40, 32), "dvr", dict(debug_hits=True, env=True, cam_pos=(0.3, -1.5, -2.0))),
    "noise32_dvr_env": ("noise", 32, (48, 40), "dvr", dict(bench=True, env=True)),
    "noise32_raymarch_env": ("noise", 32, (32, 32), "raymarch", dict(bench=True, frame=2, env=True)),
    "noise32_no_dda_env": ("noise", 32, (32, 32), "no_dda", dict(bench=True, frame=1, env=True)),
    "noise32_default_b3_env": ("noise", 32, (24, 24), "default", dict(bench=True, frame=6, bounces=3, env=True)),
}
ENV_CASES = [n for n, c in CASES.items() if c[4].get("env")]


def build_case(oracle, name):
    from tests.common import make_scene, benchmark_tf, BENCH_CAM
    from volxel_amd import synth
    from volxel_amd.transfer import default_transfer_function
    kind, n, (w, h), mode, extra = CASES[name]
    extra = dict(extra)
    if kind == "sphere":
        vox, sp = synth.sphere(n)
    else:
        vox, sp = synth.value_noise(n, seed=5, zero_quantile=0.4)
    grid = oracle.BrickGrid(vox, sp)
    frame = extra.pop("frame", 0)
    bench = extra.pop("bench", False)
    kw = dict(extra)
    if bench:
        tf, L = benchmark_tf()
        kw.update(BENCH_CAM)
        kw.setdefault("sample_range", (0.05, 1.0))
    else:
        tf, L = default_transfer_function()
    s, cam, vol, ds, p = make_scene(grid, w, h, mode, **kw)
    return grid, tf, L, p, frame


def render_case(oracle, name):
    from tests.common import default_environment
    grid, tf, L, p, frame = build_case(oracle, name)
    env = default_environment(oracle) if p.use_env else None
    return oracle.render(p, grid, tf, L, frame_index=frame, threads=1, env=env)


def tf_table_restated(colors, steps=128):
    """utils/data.ts:21-60 restated on its own (float64 like JavaScript, then Float32Array)"""
    st = sorted(colors, key=lambda c: c["stop"])
    cur, out = -1, []
    for i in range(steps):
        pos = i / steps
        if cur < 0:
            if st[0]["stop"] >= pos:
                cur = 0
                out.append(list(st[0]["color"]))
            else:
                out.append([0, 0, 0, 0])
            continue
        nxt = st[cur + 1] if cur + 1 < len(st) else None
        if nxt is None:
            out.append(list(st[cur]["color"]))
            continue
        prog = (pos - st[cur]["stop"]) / (nxt["stop"] - st[cur]["stop"])
        if prog >= 1.0:
            out.append(list(nxt["color"]))
            cur += 1
            continue
        out.append([(1 - prog) * a + prog * b for a, b in zip(st[cur]["color"], nxt["color"])])
    return np.asarray(out, dtype=np.float64).astype(np.float32).ravel()


def tf_fixture():
    """the two 128-entry tables SURVEY 8(c) asks for: viewer default ramp, benchmark.json stops"""
    from volxel_amd.settings import BENCHMARK_SETTINGS
    default = [{"color": [1, 1, 1, 0], "stop": 0}, {"color": [1, 1, 1, 1], "stop": 1}]   # viewer.ts:378-384
    return {"default": tf_table_restated(default),
            "benchmark": tf_table_restated(BENCHMARK_SETTINGS["transfer"]["transfer"]["colors"])}


def brick_fixture(oracle):
    """brick encode dumps for a 16^3 random and a 64^3 sphere volume: the three
    textures, mips, histogram gradient and a grid of Grid::lookup values"""
    from volxel_amd import synth
    out = {}
    rng = np.random.default_rng(2024)
    for tag, vox in (("r16", rng.integers(0, 4096, size=(16, 16, 16), dtype=np.uint16)), ("s64", synth.sphere(64)[0])):
        g = oracle.BrickGrid(vox)
        out[tag + "_indirection"] = g.indirection
        out[tag + "_range"] = g.range
        out[tag + "_atlas"] = g.atlas
        out[tag + "_atlas_size"] = np.array(g.atlas_size, dtype=np.uint32)
        for k in range(3):
            out[f"{tag}_mip{k}"] = g.range_mipmaps[k][0]
        vol = oracle.make_volume(g)
        n = vox.shape[0]
        pts = [(x, y, z) for z in range(0, n, max(1, n // 8)) for y in range(0, n, max(1, n // 8)) for x in range(0, n, max(1, n // 8))]
        out[tag + "_lookup"] = np.array([oracle.lib().vxo_brick_lookup(vol, x, y, z) for x, y, z in pts], dtype=np.float32)
    return out


ENV_KAT_U = [(0.0, 0.0), (0.5, 0.5), (0.25, 0.75), (0.999999, 0.000001), (0.6369617, 0.26978672),
             (0.04097353, 0.01652764), (0.8132702, 0.91275555), (0.123, 0.987)]


def env_fixture(oracle):
    """importance pyramid (levels 4..9 whole, levels 0..3 as float64 sums) and sample/lookup/pdf
    known answers of the default environment"""
    from tests.common import default_environment
    e = default_environment(oracle)
    out = {"tail": np.concatenate([e.level(k).ravel() for k in range(4, 10)]),
           "sums": np.array([e.level(k).astype(np.float64).sum() for k in range(4)]),
           "u": np.array(ENV_KAT_U, dtype=np.float32)}
    wi, lp, look, pdf = [], [], [], []
    for u0, u1 in out["u"]:
        a, b = e.sample(float(u0), float(u1), 1.5)
        wi.append(a); lp.append(b); look.append(e.lookup(a, 1.5)); pdf.append(e.pdf(a, 1.5))
    out.update(w_i=np.array(wi), le_pdf=np.array(lp), lookup=np.array(look), pdf=np.array(pdf, dtype=np.float32))
    return out


def main():
    from oracle import oracle as O
    O.build(force=True)
    for name in CASES:
        img, c = render_case(O, name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, samples=np.uint64(c.samples),
                            rays=np.uint64(c.rays), grad_samples=np.uint64(c.grad_samples))
        print(name, img.shape, "samples", c.samples, "max", float(img[..., :3].max()))
    np.savez_compressed(os.path.join(HERE, "env_default.npz"), **env_fixture(O))
    np.savez_compressed(os.path.join(HERE, "tf_tables.npz"), **tf_fixture())
    np.savez_compressed(os.path.join(HERE, "brick_dumps.npz"), **brick_fixture(O))
    L = O.lib()
    kat = {"tea": [], "wang": [], "xoshiro": []}
    for v0, v1 in [(0, 0), (1, 0), (0, 1), (42 * 12345, 17), (0xFFFFFFFF, 0xFFFFFFFF)]:
        kat["tea"].append({"v0": v0, "v1": v1, "out": int(L.vxo_tea(v0, v1, 32))})
    for x in [0, 1, 2, 3, 61, 0x12345678, 0xFFFFFFFF]:
        kat["wang"].append({"x": x, "out": int(L.vxo_wang(x))})
    for seed in [0, 1, 0xCAFEBABE]:
        raw, _ = O.rng_stream(seed, 16)
        kat["xoshiro"].append({"seed": seed, "out": [int(r) for r in raw]})
    json.dump(kat, open(os.path.join(HERE, "rng_kat.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
