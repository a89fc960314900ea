#!/bin/bash
# developer sweep of the voted kernel's knobs (run on the GPU box)
for cfg in "1 1 8" "1 1 16" "1 1 32" "2 1 8" "2 1 16" "1 2 16" "3 1 16" "1 3 32" "2 1 32" "4 1 32"; do
  set -- $cfg
  echo "vote_t=$1 vote_a=$2 k=$3: $(MI_RT_VOTE_T=$1 MI_RT_VOTE_A=$2 MI_RT_KSTEPS=$3 python tools/probe3.py 64 2>&1 | grep -E 'voted|fraction' | tr '\n' ' ')"
done
