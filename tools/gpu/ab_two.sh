#!/bin/bash
# two builds against each other with the end trims ahead, each as the FIRST context of a process of its own, alternating (tools/ab_bench.py
# on what the position does): tools/gpu/ab_two.sh a.so b.so [ab_bench options]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    PYTHONPATH=. python tools/ab_bench.py --steps 10 --rounds 3 "$@" "$L%AHEAD=1" 2>&1 | tail -1 | sed 's/total/\n    total/' | tr -s ' ' | tr '\n' ' '; echo
  done
done
