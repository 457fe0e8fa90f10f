#!/bin/bash
# usage (GPU box): tools/ab_env.sh "VAR=value" ...  -- the headline DVR kernel and the reference modes with the in-tree library under
# each environment setting ("-" = none), three passes back to back, then the image hashes (bitwise A/B)
for i in 1 2 3; do
for v in "$@"; do
  echo -n "pass $i [$v]: "
  if [ "$v" = "-" ]; then timeout -k 10 200 python tools/fpl_sweep.py --jitter 1 --fpl 32 --frames 640 2>&1 | grep 'frames/launch' | cut -c1-90
  else env "$v" timeout -k 10 200 python tools/fpl_sweep.py --jitter 1 --fpl 32 --frames 640 2>&1 | grep 'frames/launch' | cut -c1-90; fi
done; done
for v in "$@"; do
  echo "=== modes + hashes [$v]"
  if [ "$v" = "-" ]; then timeout -k 10 300 python tools/mode_probe.py --fpl 32 1 2>/dev/null | cut -c1-60; timeout -k 10 200 python tools/img_hash.py 2>/dev/null
  else env "$v" timeout -k 10 300 python tools/mode_probe.py --fpl 32 1 2>/dev/null | cut -c1-60; env "$v" timeout -k 10 200 python tools/img_hash.py 2>/dev/null; fi
done
