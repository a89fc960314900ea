#!/bin/bash
for cfg in "2 1 4" "2 1 16" "3 1 8" "3 1 16" "4 1 8" "5 2 8"; do
  set -- $cfg
  echo "vote_t=$1 vote_a=$2 k=$3: $(MI_RT_VOTE_T=$1 MI_RT_VOTE_A=$2 MI_RT_KSTEPS=$3 python tools/probe3.py 64 2>&1 | grep -E 'voted:' | tr '\n' ' ')"
done
