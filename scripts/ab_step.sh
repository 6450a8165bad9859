#!/bin/bash
# Same-box A/B of the WHOLE train step under environment switches / library builds (box-to-box spread is +-5 %).
#   scripts/ab_step.sh "NAME=ENV..." ...   e.g.  scripts/ab_step.sh "default=" "side_low=Y2_SIDE_PRIORITY=low"
# Three interleaved rounds of `bench.py` (20 event-bracketed steps + 200 sustained steps, no other legs), the arm order
# rotated every round (the first run after an idle gap reads ~0.5 % slow: an arm that always runs first looks worse than
# it is); prints the minimum of the timed and of the sustained ms per step for every arm.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/ab
arms=("$@")
for round in 1 2 3; do
  for arm in "${arms[@]}"; do
    name="${arm%%=*}"; envs="${arm#*=}"
    env $envs python bench.py --no-cpu-baseline --no-f32-mode --no-extra-legs --fed-steps 0 --sustain-steps 200 > gpurun_out/ab/step_${name}_$round.json 2> gpurun_out/ab/step_${name}_$round.err
  done
  arms=("${arms[@]:1}" "${arms[0]}")
done
python - "$@" <<'PY'
import sys, json
for a in sys.argv[1:]:
    n = a.split("=", 1)[0]
    t, s = [], []
    for r in (1, 2, 3):
        try:
            d = json.loads(open("gpurun_out/ab/step_%s_%d.json" % (n, r)).read().strip().splitlines()[-1])
            t.append(d["ms_per_step"]); s.append(d["sustained"]["ms_per_step"])
        except Exception as e:
            print(n, r, "failed:", e)
    if t:
        print("%-24s timed ms/step %s  (min %.3f)   sustained %s  (min %.3f)" % (n, " ".join("%.3f" % v for v in t), min(t), " ".join("%.3f" % v for v in s), min(s)))
PY
