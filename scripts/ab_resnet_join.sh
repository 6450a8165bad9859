#!/bin/bash
# same-box A/B of the ResNet-50 swap step (batch 32, f16): join folded into the main stack's last pass vs the two-call form
for r in 1 2 3; do
  for v in join twocall; do
    if [ $v = twocall ]; then export Y2_RESNET_TWO_CALL_JOIN=1; else unset Y2_RESNET_TWO_CALL_JOIN; fi
    for m in "--graph" ""; do
      python bench.py --model resnet50 --batch 32 $m --steps 20 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '${m:-eager}', round(d['ms_per_step'],3))"
    done
  done
done
