#!/bin/bash
# Bytes beyond L2 + launches per call of the four DTOID bench legs (separate rocprofv3 PMC passes), on the GPU box:
#   tools/collect_dtoid_traffic.sh gpurun_out/r04/pmc   ->  <dir>/dtoid_traffic.json (copy to profiles/r04_dtoid_traffic.json)
set -e
O=${1:-gpurun_out/r04/pmc}
CALLS=${2:-1}
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
for leg in forward forward_batch forward_pairs finetune; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${leg}_f -- python3 tools/dtoid_leg.py --leg $leg --calls $CALLS > $O/${leg}_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${leg}_w -- python3 tools/dtoid_leg.py --leg $leg --calls $CALLS > $O/${leg}_w.log 2>&1
  echo "$leg done"
done
python tools/pmc_dtoid_traffic.py $O $O/dtoid_traffic.json
rm -rf $O/*_f $O/*_w
