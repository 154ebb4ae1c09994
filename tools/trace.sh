# Serial (no side stream) kernel trace of a short bench run; prints the last step's per-kernel totals.
# usage (on the GPU box, from the repo root): bash tools/trace.sh [extra trace_step.py args]
R=$PWD
rm -rf $R/gpurun_out/trace
cd /tmp && export TMPDIR=/tmp CVG_NO_OVERLAP=1
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/trace -o tr --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/tr.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_step.py $f --sum > gpurun_out/trace_sum.txt
python tools/trace_step.py $f > gpurun_out/trace_all.txt
head -${1:-24} gpurun_out/trace_sum.txt; tail -1 gpurun_out/trace_sum.txt
