#!/bin/bash
cp tmp_variants/libmi_rt_base.so cs397raytracingsp22_amd/lib/libmi_rt.so
for r in 1 2 4 8 16; do TAG="rpl=$r" MI_RT_WF_TRAV_RPL=$r python tools/probe_rpl.py; done
for b in 2 4; do TAG="bpc=$b" MI_RT_WF_TRAV_BPC=$b python tools/probe_rpl.py; done
