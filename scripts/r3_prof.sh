#!/bin/bash
# Round-3 committed profiles (run from the repository root through gpurun): kernel trace of one pool of the default bench, FETCH_SIZE /
# WRITE_SIZE traffic of the engine's kernels, SQ counters of the hand-written kernels.  Condensed files land in gpurun_out/<tag>/.
tag=${1:-r3p}
R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/trace_bench.jsonl 2> $O/trace.err
python3 $R/scripts/summarize_trace.py $O/trace $O/kernel_trace.md > /dev/null
python3 $R/scripts/timeline_trace.py $O/trace $O/timeline.md > /dev/null 2>&1
python3 $R/scripts/outlier_timeline.py $O/trace $O/outlier_timeline.md 5 50 > /dev/null 2>&1
rm -rf $O/trace
rocprofv3 --pmc FETCH_SIZE -d $O/pf --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d $O/pw --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/pw.err
python3 $R/scripts/pmc_traffic.py $O/pf $O/pw 32768 $O/pmc_traffic.json > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/s1 --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/s1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -d $O/s2 --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/s2.err
mkdir -p $O/sq; cp -r $O/s1 $O/sq/; cp -r $O/s2 $O/sq/
python3 $R/scripts/summarize_pmc.py $O/sq $O/pmc_sq.md k_search k_commit k_leaf_stem k_moves k_compact k_resstage16 k_resstage32 k_convpool32 > /dev/null
rm -rf $O/pf $O/pw $O/s1 $O/s2 $O/sq
ls -la $O
