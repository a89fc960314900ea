#!/bin/bash
for pad in 0 14 27 54 110; do echo "pad=${pad}KB: $(MI_RT_LDS_PAD_KB=$pad python tools/probe3.py 64 2>&1 | grep -E 'voted|cycles' | head -2 | tr '\n' ' ')"; done
