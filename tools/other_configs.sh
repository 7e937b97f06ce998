#!/bin/bash
# GPU box: bench.py lines of the other BASELINE configurations on one GPU -> gpurun_out/<tag>_other_configs.json (one JSON object per line, keyed)
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_other_configs.jsonl
: > $OUT
run() { name=$1; shift; echo -n "{\"config\": \"$name\", \"line\": " >> $OUT; python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $OUT; echo "}" >> $OUT; }
run cfg2_bunny_720p_64spp_single            --steps 1 --warmup 1
run cfg3_tlas_720p_64spp_single             --scene tlas_scene.xml --kind 1 --steps 1 --warmup 1
run cfg3_tlas_720p_64win_job                --scene tlas_scene.xml --kind 1 --steps 64 --warmup 8
run cfg4_tower_1080p_256spp                 --scene tower_scene.xml --width 1920 --height 1080 --steps 4 --warmup 4
run cfg4_tower_1080p_64win_job              --scene tower_scene.xml --width 1920 --height 1080 --steps 64 --warmup 8
run cfg5_tlas_4k_1024spp_one_gpu            --scene tlas_scene.xml --kind 1 --width 3840 --height 2160 --steps 16 --warmup 2
echo done
