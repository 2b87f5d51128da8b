#!/bin/bash
# LU timing under different settings of the library's environment knobs (INTEGRATION.md has the table).
# Every argument is one setting: a quoted list of assignments, e.g.
#   tools/gpu_lu_sweep.sh "EIGAMD_LU_PANEL_CUS=16" "EIGAMD_LU_PANEL_CUS=64" \
#                         "EIGAMD_LU_SLICE=1024 EIGAMD_LU_SLICE_MIN=24576" "EIGAMD_LU_WIDE_T4=24576 EIGAMD_LU_WIDE_T2=16384"
# "" is the default configuration.  SIZES (default "8192 16384 32768") are the matrix orders timed per setting
# (tools/lu_time.py).  TESTS=1 runs the LU parity tests under each setting first.  One process per setting; a
# failing or hanging step ends the script.
set -o pipefail
mkdir -p gpurun_out
SIZES=${SIZES:-8192 16384 32768}
{
for setting in "${@:-}"; do
  echo "== ${setting:-defaults}"
  if [ "${TESTS:-0}" = 1 ]; then
    env $setting timeout -k 10 600 python -u -m pytest tests/test_lu_gpu.py tests/test_lu_config_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread || exit 1
  fi
  env $setting timeout -k 10 180 python -u tools/lu_time.py $SIZES || exit 1
done
} 2>&1 | tee gpurun_out/lu_sweep.log
