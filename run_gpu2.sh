#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -c "
import ctypes as C
lib = C.CDLL('srbd_horizon_amd/libsddp_hip.so')
h = C.c_void_p(); rc = lib.sddp_create(C.byref(h), 0, 30, 1, None, None); lib.sddp_last_error.restype = C.c_char_p
print('standalone (no torch) create rc', rc, lib.sddp_last_error(None))
" > gpurun_out/standalone.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t4.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
SDDP_LIB=$PWD/build/libsddp_stamps.so timeout -k 10 200 python prof_stamps.py 1024 > gpurun_out/stamps_1024.log 2>&1
SDDP_LIB=$PWD/build/libsddp_stamps.so timeout -k 10 200 python prof_stamps.py 1 > gpurun_out/stamps_1.log 2>&1
cat gpurun_out/standalone.log; grep -E "passed|failed|rc=" gpurun_out/t4.log | tail -3; tail -2 gpurun_out/smoke.log; cat gpurun_out/stamps_1024.log gpurun_out/stamps_1.log
