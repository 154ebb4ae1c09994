# Stall breakdown per kernel family (two counter passes), run from the repo root on the GPU box:
#   bash tools/pmc_stalls.sh      -> gpurun_out/stalls.json, gpurun_out/stalls.txt
R=$PWD
O=$R/gpurun_out
rm -rf $O/pmc_sa $O/pmc_sb
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU -d $O/pmc_sa -o a --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_sa.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE -d $O/pmc_sb -o b --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_sb.log 2>&1 || exit 1
cd $R
python tools/pmc_stalls.py $O/pmc_sa/a_counter_collection.csv $O/pmc_sb/b_counter_collection.csv $O/stalls.json > $O/stalls.txt || exit 1
cat $O/stalls.txt
