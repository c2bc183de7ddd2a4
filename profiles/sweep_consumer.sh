#!/bin/bash
# Frame time against the consumer grid size (MIRT_CONSUMER_BLOCKS).
for g in 512 1024 2048 4096 8192 16384 65536 262144; do
  echo -n "blocks=$g "
  MIRT_CONSUMER_BLOCKS=$g timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" || exit 1
done
