#!/bin/bash
# Phase breakdown of the fast sweep: run on the GPU box after `make -C bnmtf_amd/csrc timing`.
# Loads the timing build through BNMTF_LIB (the shipped library is not touched) and prints the per-phase cycle sums of a few blocks.
BNMTF_LIB=$(pwd)/bnmtf_amd/lib/libbnmtf_hip_timing.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" 2>&1 | grep "^block" | sort | uniq -c | sort -k3n | head -40
