#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box): kernel trace + stats, then
# one PMC pass each for FETCH_SIZE and WRITE_SIZE.  Outputs under gpurun_out/prof_bench/.
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_bench; rm -rf $out; mkdir -p $out
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --no-extra --no-cpu --no-traffic > $out/stats.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --no-extra --no-cpu --no-traffic > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --no-extra --no-cpu --no-traffic > $out/write.log 2>&1 || exit 1
cd $R
python3 scripts/pmc_traffic.py $(ls $out/fetch/*/*counter_collection.csv | head -1) $(ls $out/write/*/*counter_collection.csv | head -1) > $out/pmc_traffic.json
cat $out/pmc_traffic.json
head -5 $(ls $out/stats/*/*kernel_stats.csv | head -1) | cut -c1-200
tail -n 1 $out/stats.log | cut -c1-300
