"""Diagnostic for the spill-dependent wrong pixels of render_generic<DVR_PHONG, LAYOUT_REF>:
renders the noise32_phong fixtures with the library named by VOLXEL_HIP_LIB and prints where the image leaves the
golden image -- pixel, lane of its 8x8 wave tile, channel values -- plus the counters.
  VOLXEL_HIP_LIB=volxel_amd/libvolxel_hip_w8.so python tools/spill_probe.py [layout ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ctypes as C
    from oracle import oracle as O
    from tests.golden.make_golden import build_case
    from volxel_amd import Volxel3DRenderer
    layouts = [int(a) for a in sys.argv[1:]] or [0]
    print("library:", os.environ.get("VOLXEL_HIP_LIB", "default"))
    for name in ("noise32_phong", "noise32_phong_jitter_f2"):
        grid, tf, L, p, frame = build_case(O, name)
        want = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        for layout in layouts:
            os.environ["VX_DVR_KERNEL"] = "generic"
            r = Volxel3DRenderer(p.res[0], p.res[1], layout=layout)
            del os.environ["VX_DVR_KERNEL"]
            r.setup_from_grid(grid)
            r.change_transfer_func(tf, L)
            r.reset_counters()
            r._check(r._lib.vx_set_params(r._ctx, C.byref(p)))
            r._check(r._lib.vx_render_frame(r._ctx, frame, 0.0))
            img = r.read_accum()
            c = r.counters()
            d = np.abs(img - want["image"]).max(axis=2)
            bad = np.argwhere(d > 1e-5)
            print(f"{name} layout {layout}: max err {d.max():.3e}, {len(bad)} pixels off; samples {c.samples} "
                  f"(want {int(want['samples'])}), grads {c.grad_samples} (want {int(want['grad_samples'])})")
            for (y, x) in bad[:40]:
                # lane of the pixel in its wave tile: Morton order (vx_kernels.hpp wave_pixel)
                lx, ly = x & 7, y & 7
                lane = (lx & 1) | ((ly & 1) << 1) | ((lx & 2) << 1) | ((ly & 2) << 2) | ((lx & 4) << 2) | ((ly & 4) << 3)
                print(f"   px ({x:3d},{y:3d}) wave tile ({x >> 3},{y >> 3}) lane {lane:2d}: got {img[y, x, :3]} want {want['image'][y, x, :3]}")


if __name__ == "__main__":
    main()
