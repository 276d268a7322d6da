for cfg in "X=1" "TCVOM_NO_SCONV=1" "TCVOM_TT_NARROW=1" "TCVOM_NO_WGRADWS=1" "TCVOM_NO_SCONV=1 TCVOM_TT_NARROW=1 TCVOM_NO_WGRADWS=1"; do
  echo "== $cfg"
  for i in $(seq 1 8); do env $cfg timeout 120 python -m pytest tests/test_gpu_window.py -q -m gpu -x -s -k "window_s5_64x96" 2>&1 | grep -E "grad-norm" | sed 's/grad-norm ratio (top tensors): //' | tr "\n" ";"; done; echo
done
