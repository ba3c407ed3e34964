#!/bin/bash
set -u
mkdir -p gpurun_out/r02n
rm -f gpurun_out/r02n/ab.txt
for v in "EAP_LISTS_V2=1 EAP_LISTS_DEBUG=1" "EAP_LISTS_V2=1 EAP_LISTS_DEBUG=3" "EAP_LISTS_V2=1 EAP_LISTS_DEBUG=7"; do
  echo "== $v" >> gpurun_out/r02n/ab.txt
  env $v timeout 300 python tools/inv_pitch_experiment.py 8 2>&1 | grep "layer 2.*pitch 60" >> gpurun_out/r02n/ab.txt
done
cat gpurun_out/r02n/ab.txt
