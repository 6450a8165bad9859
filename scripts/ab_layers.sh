#!/bin/bash
# Same-box A/B of library builds / environment switches (box-to-box spread is +-5 %: never compare across gpurun calls).
#   scripts/ab_layers.sh "NAME=ENV..." ...   each arm: a label, '=', then env assignments (may be empty), e.g.
#   scripts/ab_layers.sh "new=" "r2=Y2_LIB_PATH=tensorflow_yolo2_amd/libyolo2_hip_r2.so" "bordered=Y2_HALO_COMPACT=0"
# Runs scripts/profile_layers.py for every arm, twice, interleaved; prints the per-layer fwd / dgrad / wgrad columns side by side.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/ab
for round in 1 2; do
  for arm in "$@"; do
    name="${arm%%=*}"; envs="${arm#*=}"
    env $envs python scripts/profile_layers.py > gpurun_out/ab/${name}_$round.txt 2>&1
  done
done
python - "$@" <<'PY'
import sys, re
arms=[a.split("=",1)[0] for a in sys.argv[1:]]
def load(n):
    rows={}
    for r in (1,2):
        for line in open("gpurun_out/ab/%s_%d.txt"%(n,r)):
            m=re.match(r"\s*(\d+)\s+(\d)\s+(\d+)->(\d+)\s+(\d+) \|\s+([\d.]+)\s+\d+ \|\s+([\d.]+)\s+\d+ \|\s+([\d.]+)\s+\d+ \|\s+([\d.]+)\s+([\d.]+)",line)
            if m:
                l=int(m.group(1)); v=[float(m.group(i)) for i in (6,7,8,9,10)]
                rows[l]=[min(a,b) for a,b in zip(rows.get(l,v),v)]
    return rows
data={n:load(n) for n in arms}
print("layer | "+" | ".join("%-38s"%("%s fwd dgrad wgrad bnf bnb"%n) for n in arms))
for l in sorted(data[arms[0]]):
    print("%5d | "%l+" | ".join("%7.1f %7.1f %7.1f %6.1f %6.1f "%tuple(data[n][l]) for n in arms))
print("  sum | "+" | ".join("%7.1f %7.1f %7.1f %6.1f %6.1f "%tuple(sum(data[n][l][i] for l in data[n]) for i in range(5)) for n in arms))
PY
