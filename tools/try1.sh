#!/bin/bash
for b in 8 4 3; do echo "RES bpc=$b"; MI_RT_WF_TRAV_BPC=$b python tools/probe_overlap.py 2>&1 | grep sequential | tail -2; done
