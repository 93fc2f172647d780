#!/bin/bash
# Refreshes profiles/pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes, engine source hash) for the current engine build.  usage: pmc_refresh.sh <tag>
tag=${1:-pmc}; R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/pf --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d $O/pw --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/pw.err
python3 $R/scripts/pmc_traffic.py $O/pf $O/pw 32768 $O/pmc_traffic.json > /dev/null
rm -rf $O/pf $O/pw
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
