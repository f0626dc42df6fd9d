#!/bin/bash
# Alternating A/B of two builds of the library on one box: tools/ab_lib.sh <other .so> [reps] [frames]
#   runs tools/fixed_step.py <frames> 12 with the tree's library and with SSASR_LIB=<other> in turn and prints ms/step.
R=$(cd "$(dirname "$0")/.." && pwd)
other=$1; reps=${2:-3}; frames=${3:-470}
for i in $(seq $reps); do
  out=$(timeout -k 10 200 python3 $R/tools/fixed_step.py $frames 12 2>&1 | grep 'last 3 steps' | sed 's/last 3 steps: //')
  echo "tree      $out"
  out=$(SSASR_LIB=$other timeout -k 10 200 python3 $R/tools/fixed_step.py $frames 12 2>&1 | grep 'last 3 steps' | sed 's/last 3 steps: //')
  echo "other     $out"
done
