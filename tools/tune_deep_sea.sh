#!/bin/bash
# A/B sweep of the deep_sea tile shape on one box (variants interleaved twice to expose drift).
for rep in 1 2; do
for lpb in 16 32 64 128 256; do
for u in 1 2 4 8; do
  BSX_DS_LPB=$lpb BSX_DS_UNROLL=$u python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rep=$rep lpb=$lpb unroll=$u', round(d['value']/1e9,4), 'Gsteps/s', round(d['roofline']['achieved'],1), 'GB/s', round(d['roofline']['kernel_ms'],4),'ms')"
done; done; done
