for pol in "0.5:1" "0.5:2" "0.7:1,0.4:2"; do
  echo "== narrow kernel, policy $pol"; python tools/latency_probe.py bunny_scene.xml 0 1280 720 "off;$pol" 2>&1 | sed 's/renders .* ms;/;/'
  echo "== tiles kernel only (CRT_NARROW_OFF), policy $pol"; CRT_NARROW_OFF=1 python tools/latency_probe.py bunny_scene.xml 0 1280 720 "$pol" 2>&1 | sed 's/renders .* ms;/;/'
done
