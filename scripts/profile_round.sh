#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + separate PMC passes over bench.py; raw output under
# gpurun_out/<tag>_*; scripts/summarize_profiles.py turns it into profiles/<tag>_*.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 100 --warmup 60 --no-cpu-baseline --no-unnormalised-rows --no-baseline-configs --ingest-rows 0 > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unnormalised-rows --no-baseline-configs --ingest-rows 0 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unnormalised-rows --no-baseline-configs --ingest-rows 0 > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unnormalised-rows --no-baseline-configs --ingest-rows 0 > $R/gpurun_out/${TAG}_sq.log 2>&1
# the convolutional kernels per row order (one order per process: the counters of a kernel name then belong to it) and the step over prefix levels
for ORDER in random sorted levels; do
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/${TAG}_sqcnn_${ORDER} -- python3 $R/scripts/dev/cnn_order_pmc.py ${ORDER} > $R/gpurun_out/${TAG}_sqcnn_${ORDER}.log 2>&1
done
tail -1 $R/gpurun_out/${TAG}_stats.log
