#!/bin/bash
# rocprofv3 kernel stats (min / median / average duration) of the tip:: kernels of one command.
# usage: tools/kstats.sh <tag> -- <command...>      -> gpurun_out/kstats_<tag>.txt
set -u
TAG=$1; shift; shift
export TMPDIR=/tmp
ROOT=$PWD
d=/tmp/kst_$TAG; rm -rf $d
(cd /tmp && PYTHONPATH=$ROOT timeout 900 rocprofv3 --kernel-trace --output-format csv -d $d -- "$@" > /dev/null 2>&1)
t=$(find $d -name '*kernel_trace.csv' | head -1)
mkdir -p $ROOT/gpurun_out
python $ROOT/tools/kstats_table.py "$t" | tee $ROOT/gpurun_out/kstats_$TAG.txt
