#!/bin/bash
# per-launch durations of the one-launch-per-layer dense block (csrc/dense.hip) under rocprofv3: tools/dense_trace.sh b3
B=${1:-b3}
O=gpurun_out/r04/dt
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace -d $O/t --output-format csv -- python3 tools/dense_bench.py --blocks $B --eager > $O/log.txt 2>&1
python tools/trace_list.py $O/t/*/*kernel_trace.csv --marker dense_entry --min-launches 3 > $O/list_$B.txt
rm -rf $O/t
awk '{printf "%s ", $2} END{print ""}' $O/list_$B.txt
