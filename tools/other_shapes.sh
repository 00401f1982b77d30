#!/bin/bash
# Device-resident throughput of the other BASELINE.json configurations and of small / large batches (GPU box):
#   bash tools/other_shapes.sh > gpurun_out/other_shapes.jsonl     (one bench line per configuration, --no-extras)
cd $GRAFT_REPO_ROOT
for cfg in "8 256 512" "32 384 1280" "8 448 1024" "1 1024 1920" "4 1024 1920" "8 1024 1920" "32 1024 1920" "1 320 1216" "4 2112 3840"; do
  set -- $cfg
  steps=5; warm=2
  if [ "$1" -le 1 ]; then steps=100; warm=10; fi     # a single triplet is a 1 - 2 ms step: five steps are noise
  timeout 300 python bench.py --batch $1 --height $2 --width $3 --steps $steps --warmup $warm --no-extras 2>/dev/null | grep '^{"metric"' | tail -1
done
