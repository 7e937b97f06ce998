#!/bin/bash
# GPU box: tools/latency_probe.py for every library build under build/variants/ (wide and 8-lane latency mode)
for f in $GRAFT_REPO_ROOT/build/variants/*.so; do
  echo "== $(basename $f .so)"; CRT_LIB_PATH=$f python3 $GRAFT_REPO_ROOT/tools/latency_probe.py bunny_scene.xml 0 1280 720 0,8
done
