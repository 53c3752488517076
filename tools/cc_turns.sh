#!/bin/bash
# compile kernel_sweep_turns.hip to ISA for the given slot classes and print register use:  tools/cc_ahead.sh "24 28" [-Dflags]
cd /root/repo/bnmtf_amd/csrc
ems=$1; shift
for em in $ems; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -I../../include -DTURNS_ONLY_EM=$em "$@" -S --cuda-device-only kernel_sweep_turns.hip -o /tmp/turns$em.s 2>&1 | grep -v "hip-link"
  echo "EM=$em $(grep '\.vgpr_count\|\.vgpr_spill_count' /tmp/turns$em.s | paste - - | sed -n 2p)"
done
