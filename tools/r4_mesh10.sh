#!/bin/bash
cd /root/repo
PVAMD_LIB=tools/variants/libpvamd_tune.so PYTHONPATH=tools python tools/tune_parts.py drill 500,1000,2000,3000,5000,10000,16000,20000,30000,50000,100000,150000,200000,300000,400000,520000 0 2>&1 | grep "^drill"
PVAMD_LIB=tools/variants/libpvamd_tune.so PYTHONPATH=tools python tools/tune_parts.py sphere 500,1000,2000,3000,5000,10000,16000,20000,30000,50000,100000,150000,200000,300000,400000,520000 0 2>&1 | grep "^sphere"
