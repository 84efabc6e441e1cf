#!/bin/bash
# A/B of the tile -> shard map's granularity (CT_SHARD_SHIFT: blocks of 2^k x 2^k tiles per shard): step time of rank 0 and 5 of 8.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-shard_shift}; mkdir -p "$OUT"
for k in 0 1 2 3; do
  CT_EXTRA_FLAGS=-DCT_SHARD_SHIFT=$k python -m deepestscatter_amd.build --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "== CT_SHARD_SHIFT=$k" >> "$OUT/ab.log"
  for r in 0 5; do timeout -k 10 200 python tools/shard_rank_rehearsal.py --world 8 --rank $r 2>&1 | grep '"rank"' | cut -c1-140 >> "$OUT/ab.log"; done
done
python -m deepestscatter_amd.build --force > /dev/null 2>&1
cat "$OUT/ab.log"
