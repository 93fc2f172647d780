R=$PWD; O=$R/gpurun_out/r3u; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export RP_ENGINE_LIB=$R/build_tmp/librp_engine_prev.so
rocprofv3 --pmc FETCH_SIZE -d $O/pfa --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/pfa.err
rocprofv3 --pmc WRITE_SIZE -d $O/pwa --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/pwa.err
python3 $R/scripts/pmc_traffic.py $O/pfa $O/pwa 32768 $O/pmc_prev.json > /dev/null; rm -rf $O/pfa $O/pwa
unset RP_ENGINE_LIB
rocprofv3 --pmc FETCH_SIZE -d $O/pfb --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/pfb.err
rocprofv3 --pmc WRITE_SIZE -d $O/pwb --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/pwb.err
python3 $R/scripts/pmc_traffic.py $O/pfb $O/pwb 32768 $O/pmc_cur.json > /dev/null; rm -rf $O/pfb $O/pwb
