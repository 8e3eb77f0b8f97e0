#!/bin/bash
# the WHOLE -m gpu suite with every device allocation a guard block (csrc/devmem.hip), both alignments; logs -> gpurun_out/
# usage: gpurun --timeout 3000 -- bash scripts/gpu_guard_full.sh [end|start|both] [pytest args...]
mode=${1:-both}; shift
mkdir -p gpurun_out
run() {
  VN_GUARD_ALLOC=$1 timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --deselect tests/test_gpu_guard.py "${@:2}" \
     > gpurun_out/r06_guard_full_$1.log 2>&1
  echo "guard $1: rc=$?"; tail -4 gpurun_out/r06_guard_full_$1.log
}
if [ "$mode" = both ]; then run end "$@"; run start "$@"; else run $mode "$@"; fi
