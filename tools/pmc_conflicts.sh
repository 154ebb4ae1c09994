# LDS conflict ratio of patch_fwd on single layers (one counter pass per layer): bash tools/pmc_conflicts.sh
R=$PWD
cd /tmp && export TMPDIR=/tmp
for L in "256 32 128 128" "256 8 512 512" "256 64 64 64" "256 16 256 256"; do
  tag=$(echo $L | tr ' ' '_')
  rm -rf $R/gpurun_out/pmcC_$tag
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmcC_$tag -o q --output-format csv -- python $R/tools/layer_bench.py $L 3 1 fwd 5 > $R/gpurun_out/pmcC_$tag.log 2>&1 || exit 1
  echo "== layer B H Cin Cout = $L"
  python $R/tools/pmc_show.py $R/gpurun_out/pmcC_$tag/q_counter_collection.csv
done
