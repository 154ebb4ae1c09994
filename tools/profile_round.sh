# Everything the measurement section quotes, in one GPU call (run from the repo root on the GPU box):
#   bash tools/profile_round.sh <tag>
# leaves under gpurun_out/: <tag>_kernel_stats.csv (streams overlapped, as bench.py runs),
# <tag>_kernel_stats_serial.csv (CVG_NO_OVERLAP=1: per-kernel times without co-running kernels),
# hbm_traffic.json (memory-side bytes per launch from two counter passes), mfma_util.json (matrix-pipe
# utilisation per kernel family from a third counter pass) and bench_full.json
# (the default bench line, CPU baseline included).
TAG=${1:-round}
R=$PWD
O=$R/gpurun_out
rm -rf $O/prof_o $O/prof_s $O/pmc_fetch $O/pmc_write $O/pmc_mfma
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_o -o o --output-format csv -- python $R/bench.py $ARGS > $O/prof_o.log 2>&1 || exit 1
cp $O/prof_o/o_kernel_stats.csv $O/${TAG}_kernel_stats.csv
export CVG_NO_OVERLAP=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_s -o s --output-format csv -- python $R/bench.py $ARGS > $O/prof_s.log 2>&1 || exit 1
cp $O/prof_s/s_kernel_stats.csv $O/${TAG}_kernel_stats_serial.csv
python $R/tools/trace_step.py $O/prof_s/s_kernel_trace.csv --sum > $O/${TAG}_step_serial.txt
unset CVG_NO_OVERLAP
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_mfma -o m --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_mfma.log 2>&1 || exit 1
cd $R
python tools/pmc_mfma.py $O/pmc_mfma/m_counter_collection.csv $O/mfma_util.json > $O/mfma_util.txt || exit 1
python tools/pmc_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $O/hbm_traffic.json > $O/hbm_traffic.txt || exit 1
timeout -k 10 400 python bench.py > $O/bench_full.json 2> $O/bench_full.err || exit 1
tail -1 $O/bench_full.json | cut -c1-400
