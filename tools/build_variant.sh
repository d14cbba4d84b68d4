#!/bin/bash
# A/B builds of one kernel file with extra -D flags: tools/build_variant.sh NAME FILE.hip -DX ... -> ceracoder_amd/variants/libmi355enc_NAME.so
set -e
cd "$(dirname "$0")/../ceracoder_amd/csrc"
name=$1; file=$2; shift; shift
mkdir -p ../variants
extra=""; case $file in k_deblock.hip|k_intra.hip) extra="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical $extra "$@" -c $file -o ../variants/${file%.hip}_$name.o
objs=""; for o in enc_handle.o enc_schedule.o enc_stages.o k_motion.o k_inter.o k_intra.o k_deblock.o k_handover.o h264_host.o ratecontrol.o tsmux.o; do
  if [ "$o" = "${file%.hip}.o" ]; then objs="$objs ../variants/${file%.hip}_$name.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libmi355enc_$name.so $objs -lm -lpthread
echo built ../variants/libmi355enc_$name.so
