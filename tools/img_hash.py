#!/usr/bin/env python3
"""sha256 of the accumulated image and the work counters of every render mode on a small scene: a bitwise A/B of two library
builds (VOLXEL_HIP_LIB=... python tools/img_hash.py) -- profiles/r03_bit_identity.txt"""
import os, sys, hashlib
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
r, msg, info = bench.build_scene(640, 360, 128, 0, 1, 0)
for mode, b in (("raymarch", 1), ("raymarch", 3), ("default", 2), ("no_dda", 2), ("dvr", 1), ("dvr_phong", 1)):
    r.settings.render_mode, r.settings.bounces = mode, b
    r.restart_rendering(); r.bind_uniforms(); r.reset_counters()
    r.render(frames=int(os.environ.get("IMG_HASH_FRAMES", "8")), rebind=False, in_flight=int(os.environ.get("IMG_HASH_INFLIGHT", "4"))); r.finish()
    img = r.read_accum(); c = r.counters()
    print(mode, b, hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()[:16], c.samples, c.tf_samples, c.skip_steps, flush=True)
