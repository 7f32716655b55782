#!/bin/bash
for mode in default eager warm; do
  case $mode in
    default) env="";;
    eager) env="HIP_ENABLE_DEFERRED_LOADING=0";;
    warm) env="PVAMD_STALL_WARMUP=1";;
  esac
  echo "== $mode ($env)"
  env $env timeout 300 python tools/stall_trace.py > /tmp/stall_$mode.txt 2>&1
  grep "steps above\|Error\|error" /tmp/stall_$mode.txt | cut -c1-700
done
