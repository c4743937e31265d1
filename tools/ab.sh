#!/bin/bash
# same-box A/B of two prebuilt libraries: tools/ab.sh "<command>"   (lib/base.so vs lib/new.so, alternated twice)
L=multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd/lib
for v in base new base new; do cp $L/$v.so $L/libavformer_hip.so; echo "== $v"; eval "$1"; done
