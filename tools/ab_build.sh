#!/bin/bash
# A/B of a compile-time switch on the GPU box: tools/ab_build.sh "<extra hipcc flags>" <tag>
# rebuilds the library with the flags, then runs the per-layer and whole-leg benches into gpurun_out/ab_<tag>.*
set -e
mkdir -p gpurun_out
OSSID_HIPCC_EXTRA="$1" python -c "from ossid_code_amd import _build; _build.build_lib(force=True)"
python tools/train_layers_bench.py --what fwd,dgrad > gpurun_out/ab_$2.layers.txt 2>&1
python tools/bench_finetune.py --reps 8 > gpurun_out/ab_$2.finetune.txt 2>&1
python tools/bench_dtoid.py --what forward > gpurun_out/ab_$2.forward.txt 2>&1
tail -1 gpurun_out/ab_$2.layers.txt; tail -2 gpurun_out/ab_$2.finetune.txt; tail -2 gpurun_out/ab_$2.forward.txt
