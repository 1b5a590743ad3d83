#!/bin/bash
# A/B of the whole headline step under an environment switch, interleaved to cancel box drift.
# usage: tools/ab_step.sh VAR val_a val_b [reps]   (prints ms_per_step of `bench.py --no-extra --no-cpu-baseline --steps 300`)
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for v in $A $B; do
    ms=$(env $VAR=$v python bench.py --no-extra --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$VAR=$v ms_per_step=$ms"
  done
done
