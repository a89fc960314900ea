#!/bin/bash
python tools/probe3.py 256 cfg1 2>&1 | grep -E "kernel_ms"
python tools/probe3.py 256 cfg4 2>&1 | grep -E "kernel_ms"
python tools/probe3.py 256 cfg5 2>&1 | grep -E "kernel_ms"
python tools/probe3.py 64 head 2>&1 | grep -E "kernel_ms"
