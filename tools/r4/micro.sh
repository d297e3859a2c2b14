#!/bin/bash
O=gpurun_out/r4_micro; mkdir -p $O
./tools/micro/valu_issue.bin > $O/valu_issue.txt 2>&1; echo "micro rc=$?"
cat $O/valu_issue.txt
