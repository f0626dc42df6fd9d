#!/bin/bash
# Alternating A/B of one environment switch on one box: tools/ab_env.sh NAME A B [reps] [frames]
#   runs tools/fixed_step.py <frames> 9 with NAME=A and NAME=B in turn, `reps` times, and prints ms/step.
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; a=$2; b=$3; reps=${4:-3}; frames=${5:-470}
for i in $(seq $reps); do
  for v in $a $b; do
    out=$(env $name=$v timeout -k 10 200 python3 $R/tools/fixed_step.py $frames 9 2>&1 | grep 'last 3 steps' | sed 's/last 3 steps: //')
    echo "$name=$v  $out"
  done
done
