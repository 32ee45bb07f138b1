#!/bin/bash
# A/B of convolution schedules / builds on ONE box: tools/conv_shapes.py (one train-mode forward+backward at the BASELINE
# size) once per setting.  Usage: tools/ab_conv_shapes.sh <out-prefix> VAR=value [VAR=value ...]   (one run per setting,
# plus one run per library variant found under tools/_ab_<name>/libonda_hip.so, built by hand from another revision)
out=$1; shift
python tools/conv_shapes.py > ${out}_base.txt 2>&1
for kv in "$@"; do env $kv python tools/conv_shapes.py > ${out}_${kv//[^A-Za-z0-9]/_}.txt 2>&1; done
for v in tools/_ab_*/; do
  n=$(basename $v); n=${n#_ab_}
  [ -f $v/libonda_hip.so ] && ONDA_LIB_PATH=$PWD/$v/libonda_hip.so python tools/conv_shapes.py > ${out}_lib_$n.txt 2>&1
done
grep -H "total conv" ${out}_*.txt
