#!/bin/bash
# all BASELINE configs at 1080p, reduced spp where noted (developer probe, run on the GPU box)
python tools/probe3.py 256 cfg1 2>&1 | grep -E "kernel_ms"
python tools/probe3.py 256 cfg4 2>&1 | grep -E "kernel_ms|cycles"
python tools/probe3.py 256 cfg5 2>&1 | grep -E "kernel_ms|cycles"
python tools/probe3.py 64 head 2>&1 | grep -E "kernel_ms|cycles"
