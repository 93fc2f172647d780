#!/bin/bash
# c5 (50x50 / 128 / 800): bounded window of the bench, then a kernel trace of a shorter window.  usage: r3_c5.sh <tag>
tag=${1:-r3e}; R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
python bench.py --config c5 --waves 600 --no-cpu-baseline > $O/c5_window.json 2> $O/c5_window.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --config c5 --waves 150 --no-cpu-baseline > $O/c5_trace_bench.json 2> $O/c5_trace.err
python3 $R/scripts/summarize_trace.py $O/trace $O/c5_kernel_trace.md > /dev/null; rm -rf $O/trace
cd $R; tail -c 900 $O/c5_window.json; head -30 $O/c5_kernel_trace.md | cut -c1-200
