#!/bin/bash
# round-3 debugging: one fuzz option set on the GPU through the run-time compiled kernel
O=gpurun_out/r3_dbg; mkdir -p $O
run() { python tools/fuzz_parity.py gpu 1 4096 scalars=1 idveg=4 iopt_run=3 iopt_inf=1 iopt_frz=2 --seed:1 > $O/$1.log 2>&1; echo "== $1: $(tail -n 1 $O/$1.log)"; }
run base
