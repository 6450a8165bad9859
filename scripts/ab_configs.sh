#!/bin/bash
# Same-box A/B of two library builds on configs[0] / [1] / [2] (C1 single-image latency, C2 core forward, C3 classifier
# step): scripts/ab_configs.sh "NAME=ENV..." ...  -- two interleaved rounds, minimum of each.  Box-to-box spread is +-4 %.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/ab
for round in 1 2; do
  for arm in "$@"; do
    name="${arm%%=*}"; envs="${arm#*=}"
    env $envs python bench.py --forward-only --batch 32 --no-cpu-baseline --no-f32-mode --kernel-events off --sustain-steps 0 > gpurun_out/ab/c2_${name}_$round.json 2>/dev/null
    env $envs python bench.py --model classifier > gpurun_out/ab/c3_${name}_$round.json 2>/dev/null
    env $envs BATCH=1 SIZE=224 python scripts/profile_forward_layers.py 2>/dev/null | tail -1 > gpurun_out/ab/c1_${name}_$round.txt
  done
done
python - "$@" <<'PY'
import sys, json, re
for a in sys.argv[1:]:
    n = a.split("=", 1)[0]
    c2 = [json.loads(open("gpurun_out/ab/c2_%s_%d.json" % (n, r)).read().strip().splitlines()[-1])["ms_per_step"] for r in (1, 2)]
    c3 = [json.loads(open("gpurun_out/ab/c3_%s_%d.json" % (n, r)).read().strip().splitlines()[-1])["ms_per_step"] for r in (1, 2)]
    c1 = [float(re.search(r"total us ([\d.]+)", open("gpurun_out/ab/c1_%s_%d.txt" % (n, r)).read()).group(1)) for r in (1, 2)]
    print("%-6s C2 core forward 416^2 x32: %.3f ms   C3 classifier step 224^2 x128: %.3f ms   C1 one 224^2 image, serialised launches: %.0f us"
          % (n, min(c2), min(c3), min(c1)))
PY
