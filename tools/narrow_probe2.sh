for st in 8 1; do
  echo "== tile stride $st, narrow kernel on"; env PROBE_TILE_STRIDE=$st CRT_LAT_VERBOSE=1 python tools/latency_probe.py bunny_scene.xml 0 1280 720 "default" 2>gpurun_out/r03/p2.err | sed 's/renders .* ms;/;/'; grep "latency stages" gpurun_out/r03/p2.err | tail -1
done
