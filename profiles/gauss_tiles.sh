#!/bin/bash
# single-pass Gaussian (reg step, 5 taps) on a C3-sized volume: tile shapes and z chunk lengths
for t in 64x16 64x32 128x8 128x16; do
  for z in 128 256 512; do
    echo -n "tile $t zchunk $z: "; MI_GAUSS_TILE=$t MI_GAUSS_ZCHUNK=$z python3 profiles/gauss_time.py 2>/dev/null | head -n 1
  done
done
