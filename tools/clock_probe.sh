#!/bin/bash
# Lab: shader clock / power while a kernel loop runs (is the split-core GEMM's k-loop clock-limited?).
# usage: tools/clock_probe.sh <label> <command...>   -- samples rocm-smi every 0.2 s while the command runs
label=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
sleep 1.5
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' ' | sed "s/^/$label: /"
  echo
  sleep 0.25
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
