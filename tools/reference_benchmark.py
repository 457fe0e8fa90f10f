#!/usr/bin/env python3
"""The reference's own benchmark scenario (public/benchmark.json: default / no_dda / raymarch, shared
settings 0: 500 samples, bounces 1, resolutionFactor 0.8, environment lighting) through
Volxel3DRenderer.start_benchmark on the config-3 volume.  Prints the VolxelBenchmarkResult records'
timePerSample (ms, per-frame draw + finish like viewer.ts:1213-1218)."""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volxel_amd import (BENCHMARK_SETTINGS, BENCHMARK_COLLECTION_MODES, Volxel3DRenderer, read_u16_stack_to_grid, synth)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vox, sp = synth.value_noise(n, seed=42)
r = Volxel3DRenderer(1920, 1080)
r.setup_from_grid(read_u16_stack_to_grid(vox, sp))
coll = {"sharedSettings": [copy.deepcopy(BENCHMARK_SETTINGS)],
        "benchmarks": [{"renderMode": m, "settings": 0, "name": m} for m in BENCHMARK_COLLECTION_MODES]}
r.start_benchmark(coll)                       # warm-up pass (allocations, launch orders)
for rec in r.start_benchmark(coll):
    print(json.dumps({"name": rec["name"], "timePerSample_ms": round(rec["timePerSample"], 4),
                      "totalTime_ms": round(rec["totalTime"], 1), "viewport": rec["viewport"],
                      "samples": rec["settings"]["maxSamples"]}), flush=True)
