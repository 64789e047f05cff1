#!/bin/bash
# gpurun_out/r05e/* (tools/profile_r05.sh a, b, c) -> profiles/r05_*
R=gpurun_out/r05e
for f in $(ls $R | grep -v "progress_\|\.err$\|\.log$"); do cp $R/$f profiles/r05_$f; done
[ -f $R/traffic.json ] && cp $R/traffic.json profiles/traffic.json
