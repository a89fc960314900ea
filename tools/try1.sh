#!/bin/bash
for m in 3 0; do echo "RES mode=$m"; MI_RT_WF_TRAV_LDS=$m python tools/probe_cfgs.py cfg4; done
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
