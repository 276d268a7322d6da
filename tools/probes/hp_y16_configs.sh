for cfg in "A|" "B|TCVOM_Y16_LAYERS=layer2" "C|TCVOM_HP_LAYERS=conv1,conv2,conv3 TCVOM_Y16_LAYERS=layer1,layer2" "D|TCVOM_HP_LAYERS=conv1,conv2,conv3 TCVOM_Y16_LAYERS=layer1,layer2,layer3"; do
  name=${cfg%%|*}; envs=${cfg#*|}
  echo "== $name: $envs"
  env $envs timeout 600 python -m pytest tests/test_gpu_window.py -q -s -k "north_star_parity or test_window_256" 2>&1 | grep -E "unknown-only|passed|failed" | sed -E 's/, [0-9]+ unknown pixels.*//; s/ ; losses.*//' | cut -c1-120
  env $envs python bench.py --steps 12 --no-cpu-baseline --no-profile --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; print('ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"
done
