#!/bin/bash
# Development build with wall-clock marks in the P-picture kernels (-DTL_PROF): ceracoder_amd/variants/libmi355enc_TL.so, read by tests/devtools/timeline.py
set -e
cd "$(dirname "$0")/../ceracoder_amd/csrc"
mkdir -p ../variants
for f in k_motion k_intra k_deblock; do
  extra=""; case $f in k_deblock|k_intra) extra="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical $extra -DTL_PROF -c $f.hip -o ../variants/${f}_TL.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libmi355enc_TL.so enc_handle.o enc_schedule.o enc_stages.o ../variants/k_motion_TL.o k_inter.o ../variants/k_intra_TL.o ../variants/k_deblock_TL.o k_handover.o h264_host.o ratecontrol.o tsmux.o -lm -lpthread
echo built ../variants/libmi355enc_TL.so
