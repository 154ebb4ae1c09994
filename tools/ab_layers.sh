#!/bin/bash
# A/B of the stride-1 128-channel-tile layers of the c3 step, one line per layer and mode.
#   tools/ab_layers.sh "ENV_A=..." "ENV_B=..."      (each argument: environment assignments for one arm)
#   LAYERS="256 32 128 128 3 1;256 16 256 256 3 1" tools/ab_layers.sh ...   (subset)
set -e
all="256 32 128 128 3 1;256 16 256 256 3 1;256 8 512 512 3 1;256 8 512 512 3 0;256 16 512 256 3 0;256 32 256 128 3 0;256 66 64 128 3 0;256 34 128 256 3 0;256 18 256 512 3 0;256 10 512 512 3 0"
IFS=';' read -ra layers <<< "${LAYERS:-$all}"
for arm in "$@"; do
  echo "== $arm"
  for l in "${layers[@]}"; do
    env $arm timeout -k 10 120 python tools/layer_bench.py $l fwd 50 2>&1 | grep -v amdgpu.ids
  done
done
