#!/bin/bash
# Phase breakdown of the fast sweep: run on the GPU box after `make -C bnmtf_amd/csrc timing`.
# Swaps the timing build in (on the box copy only) and prints the per-phase cycle sums of a few blocks.
cp bnmtf_amd/lib/libbnmtf_hip_timing.so bnmtf_amd/lib/libbnmtf_hip.so
python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" 2>&1 | grep "^block" | sort | uniq -c | sort -k3n | head -40
