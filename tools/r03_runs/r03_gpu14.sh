#!/bin/bash
O=gpurun_out/r03n
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/host_split.py 2>&1 | grep -v Warn | tail -12 > $O/host_split.log
timeout 200 python tools/host_split.py --extra-hparams wn_stack_executor=False,defer_wgrad_reduce=False 2>&1 | grep -v Warn | tail -12 > $O/host_split_old.log
cat $O/host_split.log $O/host_split_old.log
