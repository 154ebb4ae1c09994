# Memory-side bytes per launch of every kernel family for one configuration (two counter passes), written where
# bench.py --config <config> picks it up:  bash tools/traffic_config.sh <tag> <config>   (GPU box, repo root)
# leaves gpurun_out/<tag>_hbm_traffic_<config>.json (copy it to profiles/) and <tag>_kernel_stats_<config>.csv.
TAG=${1:-round}; CFG=${2:-c2}
R=$PWD; O=$R/gpurun_out
rm -rf $O/pmc_fetch_$CFG $O/pmc_write_$CFG $O/prof_$CFG
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$CFG -o f --output-format csv -- python $R/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_fetch_$CFG.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$CFG -o w --output-format csv -- python $R/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_write_$CFG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$CFG -o p --output-format csv -- python $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_$CFG.log 2>&1 || exit 1
cp $O/prof_$CFG/p_kernel_stats.csv $O/${TAG}_kernel_stats_$CFG.csv
cd $R
python tools/pmc_traffic.py $O/pmc_fetch_$CFG/f_counter_collection.csv $O/pmc_write_$CFG/w_counter_collection.csv $O/${TAG}_hbm_traffic_$CFG.json > $O/${TAG}_hbm_traffic_$CFG.txt || exit 1
grep patch_fwd $O/${TAG}_hbm_traffic_$CFG.txt
