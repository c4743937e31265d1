#!/bin/bash
# rebuild lib/libavformer_hip.so (and the stamped diagnostic build with --dbg) from any directory
set -e
HERE=$(cd "$(dirname "$0")/../.." && pwd)
cd "$HERE"
python - <<'PY'
import sys
sys.path.insert(0, '.')
import avformer_amd  # noqa: F401  (registers the package alias)
from importlib import import_module
b = import_module('multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd._build')
b.build(verbose=True)
PY
if [ "$1" == "--dbg" ]; then tools/diag/build_ws_dbg.sh; fi
