#!/bin/bash
# Collects the evidence files of a round on the GPU box into gpurun_out/<tag>/ (copied to profiles/ afterwards):
#   tools/collect_round_evidence.sh r02
set -e
T=${1:-r04}
O=gpurun_out/$T/ev
mkdir -p $O
R=$PWD
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 > $O/bench_under_profiler.json 2> $O/stats.err
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
python tools/stats_by_pass.py $O/stats/*/*kernel_trace.csv --steps 20 --warmup 5 > $O/kernel_stats_by_pass.txt
rocprofv3 --kernel-trace -d $O/ft --output-format csv -- python3 tools/bench_finetune.py --reps 3 --no-graph > $O/ft.log 2>&1
python tools/trace_step.py $O/ft/*/*kernel_trace.csv --top 45 > $O/finetune_step_kernels.txt
rocprofv3 --kernel-trace -d $O/fw --output-format csv -- python3 tools/bench_dtoid.py --what forward > $O/fw.log 2>&1
python tools/trace_step.py $O/fw/*/*kernel_trace.csv --marker detect_emit --top 40 > $O/forward_step_kernels.txt
python tools/train_layers_bench.py --json $O/train_layers.json > $O/train_layers.txt 2>&1
python tools/train_layers_bench.py --what fwd,dgrad --wino > $O/train_layers_wino.txt 2>&1
for k in fwd_x6 fwd_exact; do python tools/train_layers_bench.py --what fwd --fwd-kind $k --only "^b[1-4]|^t[1-3]" > $O/train_layers_$k.txt 2>&1; done
python tools/train_layers_bench.py --what fwd --wino --batch 21 --only . > $O/forward_layers_wino_nt21.txt 2>&1
rm -rf $O/stats $O/ft $O/fw
ls -la $O
python tools/stem_bench.py > $O/stem_kernels.txt 2>&1
python tools/wgrad_group_bench.py > $O/wgrad_group.txt 2>&1
rocprofv3 --kernel-trace -d $O/ft2 --output-format csv -- python3 tools/bench_finetune.py --reps 3 --no-graph > $O/ft2.log 2>&1
python tools/step_timeline.py $O/ft2/*/*kernel_trace.csv > $O/finetune_step_timeline.txt
python tools/step_gaps.py $O/ft2/*/*kernel_trace.csv --min-us 100 > $O/finetune_step_gaps.txt
rm -rf $O/ft2
python tools/dense_bench.py --blocks b1,b2,b3,b4 > $O/dense_blocks.txt 2>&1
python tools/forward_split.py > $O/forward_split.txt 2>&1
rocprofv3 --kernel-trace -d $O/fw2 --output-format csv -- python3 tools/bench_dtoid.py --what forward > $O/fw2.log 2>&1
python tools/trace_list.py $O/fw2/*/*kernel_trace.csv --marker detect_emit > $O/forward_step_list.txt
rm -rf $O/fw2
python tools/bench_finetune.py --reps 10 --no-graph > $O/finetune_ms.txt 2>&1
python tools/step_phases.py > $O/finetune_step_phases.txt 2>&1
rocprofv3 --kernel-trace -d $O/ft3 --output-format csv -- python3 tools/bench_finetune.py --reps 3 --no-graph > $O/ft3.log 2>&1
python tools/trace_list.py $O/ft3/*/*kernel_trace.csv > $O/finetune_step_list.txt
rm -rf $O/ft3
# traffic beyond L2 + launches of the finetune leg (steady state: the third call)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/finetune_f -- python3 tools/dtoid_leg.py --leg finetune --calls 3 > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/finetune_w -- python3 tools/dtoid_leg.py --leg finetune --calls 3 > $O/pmc_w.log 2>&1
python tools/pmc_dtoid_traffic.py $O/pmc $O/finetune_traffic.json > $O/finetune_traffic.txt 2>&1
rm -rf $O/pmc
ls -la $O
