#!/bin/bash
# tools/ab_layers.sh "<extra hipcc flags>" <tag> [train_layers_bench args]: rebuild with the flags, per-layer rates only
set -e
mkdir -p gpurun_out
FL="$1"; TAG="$2"; shift 2
OSSID_HIPCC_EXTRA="$FL" python -c "from ossid_code_amd import _build; _build.build_lib(force=True)"
python tools/train_layers_bench.py "$@" > gpurun_out/ab_$TAG.layers.txt 2>&1
echo "== $TAG"; grep -v amdgpu.ids gpurun_out/ab_$TAG.layers.txt | tail -14
