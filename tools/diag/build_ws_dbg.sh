#!/bin/bash
# Diagnostic build of the library with in-kernel stamps in the weight-stationary GEMM: lib/ws_dbg.so (AVF_LIB_PATH).
# Only gemm_ws.hip is recompiled (-DAVF_WS_STAMPS=1); the other objects come from the product build (run _build.py first).
set -e
HERE=$(cd "$(dirname "$0")/../.." && pwd)
PK="$HERE/multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd"
mkdir -p "$PK/lib/obj_dbg"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAVF_WS_STAMPS=1 ${AVF_WS_DBG_DEFS} -c "$PK/csrc/gemm_ws.hip" -o "$PK/lib/obj_dbg/gemm_ws.o"
OBJS=$(ls "$PK"/lib/obj/*.o | grep -v gemm_ws.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PK/lib/ws_dbg.so" $OBJS "$PK/lib/obj_dbg/gemm_ws.o"
echo "$PK/lib/ws_dbg.so"
