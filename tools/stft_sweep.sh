#!/bin/bash
# Builds variants of csrc/stft.hip alone (tuning knobs SAT_STFT_NG / SAT_STFT_FBPTS / SAT_STFT_CCMAX) into tools/_stft_variants/<tag>.so (the other
# objects come from csrc/build/) — run here, where hipcc cross-compiles; then on the GPU box: python tools/stft_bench.py --lib tools/_stft_variants/<tag>.so
set -e
cd "$(dirname "$0")/../stable_audio_tools_amd/csrc"
make -s all
OUT=../../tools/_stft_variants; mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -fno-slp-vectorize"
OTHERS=$(ls build/*.o | grep -v stft.o)
for v in "ng8_fb512_cc1:-DSAT_STFT_NG=8" "ng4_fb512_cc1:-DSAT_STFT_NG=4" "ng8_fb1024_cc1:-DSAT_STFT_FBPTS=1024" "ng4_fb1024_cc1:-DSAT_STFT_NG=4 -DSAT_STFT_FBPTS=1024" "ng8_fb512_cc2:-DSAT_STFT_CCMAX=2" "ng2_fb1024_cc1:-DSAT_STFT_NG=2 -DSAT_STFT_FBPTS=1024"; do
  tag=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c stft.hip -o $OUT/stft_$tag.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS $OUT/stft_$tag.o -o $OUT/$tag.so
  rm $OUT/stft_$tag.o
  echo built $tag
done
