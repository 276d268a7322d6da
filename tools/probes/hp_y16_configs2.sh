for cfg in "A|" "B|TCVOM_Y16_LAYERS=layer2" "C|TCVOM_HP_LAYERS=conv1,conv2,conv3 TCVOM_Y16_LAYERS=layer1,layer2" "E|TCVOM_Y16_LAYERS=layer2,layer3,layer_bottleneck"; do
  name=${cfg%%|*}; envs=${cfg#*|}
  echo "== $name: $envs"
  for i in 1 2; do env $envs timeout 600 python -m pytest tests/test_gpu_window.py -q -s -k "large_vs_oracle" 2>&1 | grep -E "unknown-only|passed|failed" | sed -E 's/, spatial.*//' | cut -c1-120; done
done
