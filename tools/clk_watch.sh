#!/bin/bash
# Samples GPU clock / power while a GEMM micro-benchmark loop runs (is the MFMA loop clock-throttled?).
# usage: tools/clk_watch.sh <variant> [iters]
V=${1:-2}; IT=${2:-400}
OUT=gpurun_out/clk_watch_v$V.txt
mkdir -p gpurun_out
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > $OUT &
SMI=$!
sleep 1.5
PGIBBS_BENCH_ITERS=$IT python tools/gemm_bench.py $V
wait $SMI
cat $OUT
