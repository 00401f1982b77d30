#!/bin/bash
# ONE command that turns `parity: unpinned` into a reference-held pin, on any machine with Torch7 (th) + cutorch / cunn / cudnn / nngraph /
# image + the reference checkout with stn and spy built (README.md:23-33 of the reference) and, for the second half, the pretrained weights:
#
#     bash tools/pin_from_torch7.sh /path/to/back2future [Ours-Soft-ft-KITTI]
#
# 1. dump_ops.lua   (no weights needed): every third-party op the oracle restates, on counter-generated tensors, the whole output table of a fresh
#    createModelMulti() and that model as a Torch-written tiny_model.t7            -> tests/golden/torch7_dump/ops/
# 2. dump_samples.lua (needs models/*.t7): computeFlow on the reference's samples/ triplet + a copy of the .t7 -> tests/golden/torch7_dump/
# 3. python -m pytest tests/test_reference_dump.py -q      (the oracle on the CPU; with a GPU also the library, .t7 reader included)
# Commit tests/golden/torch7_dump/ afterwards: test_reference_dump.py stops skipping for everyone.
set -euo pipefail
REF=${1:?usage: bash tools/pin_from_torch7.sh /path/to/back2future [model name]}
NAME=${2:-Ours-Soft-ft-KITTI}
HERE=$(cd "$(dirname "$0")/.." && pwd)
D=$HERE/tests/golden/torch7_dump
command -v th >/dev/null || { echo "th (Torch7) not found on PATH" >&2; exit 2; }
mkdir -p "$D/ops"
(cd "$REF" && th "$D/dump_ops.lua" "$D/ops")
if ls "$REF"/models/*.t7 >/dev/null 2>&1; then
  (cd "$REF" && th "$D/dump_samples.lua" "$NAME" "$D")
  # the file back2future.init(name) loads (back2future.lua:100-110) travels with the dump as weights.t7
  case "$NAME" in
    Ours-Hard) W=models/RoamingImages_H.t7 ;;
    Ours-Soft-ft-KITTI) W=models/RoamingImages_H_KITTI_S.t7 ;;
    Ours-Soft-ft-Sintel) W=models/RoamingImages_H_Sintel_S.t7 ;;
    *) W= ;;
  esac
  if [ -n "$W" ] && [ -f "$REF/$W" ]; then cp "$REF/$W" "$D/weights.t7"; else echo "copy the .t7 that '$NAME' maps to as $D/weights.t7 by hand" >&2; fi
else
  echo "no models/*.t7 in $REF: skipped dump_samples.lua (the op dump alone already pins every [3P] op and the .t7 reader)" >&2
fi
cd "$HERE" && python -m pytest tests/test_reference_dump.py -q
