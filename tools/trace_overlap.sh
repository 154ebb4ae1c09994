# Kernel trace of a short bench run as bench.py schedules it (side streams on); leaves the last step's
# dispatch list in gpurun_out/trace_ov_all.txt and an idle-gap summary in gpurun_out/trace_ov_gaps.txt.
# usage (on the GPU box, from the repo root): bash tools/trace_overlap.sh
R=$PWD
rm -rf $R/gpurun_out/trace_ov
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_ov -o tr --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/tr_ov.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/trace_ov -name "*kernel_trace.csv" | head -1)
python tools/trace_step.py $f > gpurun_out/trace_ov_all.txt
python tools/trace_gaps.py $f > gpurun_out/trace_ov_gaps.txt
tail -25 gpurun_out/trace_ov_gaps.txt
