#!/bin/bash
# A/B of developer knobs (MI_RT_* environment variables, read once at mi_ctx_create) on ONE box, one process per setting:
#   tools/ab_env.sh "<probe.py args>" "VAR1=a VAR2=b" "VAR1=c" ...
ARGS="$1"; shift
for rep in 1 2; do for e in "$@"; do echo "RES [$e] $(env $e python tools/probe.py $ARGS 2>&1 | grep -a '^RES' | head -1)"; done; done
