#!/bin/bash
# A/B of convolution kernel builds on ONE box: tools/conv_shapes.py (one train-mode forward+backward at the BASELINE size)
# per library variant (tools/_ab_<name>/libonda_hip.so, built by hand from another revision of csrc/conv_l2.hip) and per
# schedule knob of the in-tree library.  Usage: tools/ab_conv_shapes.sh <out-prefix>
out=${1:-gpurun_out/ab}
for v in tools/_ab_*/; do
  n=$(basename $v); n=${n#_ab_}
  [ -f $v/libonda_hip.so ] && ONDA_LIB_PATH=$PWD/$v/libonda_hip.so python tools/conv_shapes.py > ${out}_$n.txt 2>&1
done
for xt in 1 3 4; do ONDA_L2_XT=$xt python tools/conv_shapes.py > ${out}_tree_xt$xt.txt 2>&1; done
head -2 ${out}_*.txt
