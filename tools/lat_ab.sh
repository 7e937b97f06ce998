#!/bin/bash
# GPU box: tools/latency_probe.py (one wave per tile, and the tuned latency mode) for every library build under build/variants/
for f in $GRAFT_REPO_ROOT/build/variants/*.so; do
  echo "== $(basename $f .so)"; CRT_LIB_PATH=$f python3 $GRAFT_REPO_ROOT/tools/latency_probe.py bunny_scene.xml 0 1280 720 "off;default" | cut -c 1-175
done
