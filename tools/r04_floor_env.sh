#!/bin/bash
# Runs on the GPU box: the launch-floor probe under a few runtime settings (env is set BEFORE rocprofv3 starts the program).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "default" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=1"; do
  tag=$(echo "$cfg" | tr '=' '_')
  if [ "$cfg" != "default" ]; then export "$cfg"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04/floor_$tag -- $R/tools/probes/launch_floor > /dev/null 2>&1 || echo "fail $cfg"
  if [ "$cfg" != "default" ]; then unset "${cfg%%=*}"; fi
done
echo ok
