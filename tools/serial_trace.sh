# Per-kernel totals of one step on one stream: bash tools/serial_trace.sh <tag> [kernel-name filter for a per-dispatch list]
# (run from the repo root on the GPU box)
TAG=${1:-serial}
R=$PWD; O=$R/gpurun_out
rm -rf $O/prof_s
cd /tmp && export TMPDIR=/tmp
export CVG_NO_OVERLAP=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_s -o s --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_s.log 2>&1 || exit 1
python $R/tools/trace_step.py $O/prof_s/s_kernel_trace.csv --sum > $O/${TAG}_step_serial.txt
python $R/tools/trace_step.py $O/prof_s/s_kernel_trace.csv > $O/${TAG}_step_dispatches.txt
cp $O/prof_s/s_kernel_stats.csv $O/${TAG}_kernel_stats_serial.csv
rm -rf $O/prof_s
if [ -n "$2" ]; then grep -E "$2" $O/${TAG}_step_dispatches.txt; else cat $O/${TAG}_step_serial.txt; fi
