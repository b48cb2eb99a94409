#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of EVERY BASELINE config's own optimizer step (scripts/baseline_configs.py configsK: the
# host drivers' loop, one config per process so that a kernel name's launches belong to one config), raw output under
# gpurun_out/<tag>_cfgK; scripts/summarize_configs.py <tag> turns it into profiles/<tag>_configs_summary.md.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for K in 1 2 4 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cfg${K} -- python3 $R/scripts/baseline_configs.py configs${K} > $R/gpurun_out/${TAG}_cfg${K}.log 2>&1 || exit 1
done
