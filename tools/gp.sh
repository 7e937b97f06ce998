#!/bin/bash
# authoring container: run a command on the GPU box with gpurun_out/r03 created first.  usage: tools/gp.sh [timeout] 'command'
T=900; if [[ "$1" =~ ^[0-9]+$ ]]; then T=$1; shift; fi
/usr/local/graft/bin/gpurun --timeout $T -- "mkdir -p gpurun_out/r03 && $*" 2>&1 | grep -v "^\[gpurun\] sending\|merged"
