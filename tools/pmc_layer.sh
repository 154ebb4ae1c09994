# Two counter passes over one layer benchmark: bash tools/pmc_layer.sh <layer_bench args>
# (on the GPU box, from the repo root).  Results: gpurun_out/pmcA, gpurun_out/pmcB.
R=$PWD
rm -rf $R/gpurun_out/pmcA $R/gpurun_out/pmcB
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmcA -o q --output-format csv -- python $R/tools/layer_bench.py "$@" > $R/gpurun_out/pmcA.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum -d $R/gpurun_out/pmcB -o q --output-format csv -- python $R/tools/layer_bench.py "$@" > $R/gpurun_out/pmcB.log 2>&1 || exit 1
cd $R && python tools/pmc_show.py gpurun_out/pmcA/q_counter_collection.csv gpurun_out/pmcB/q_counter_collection.csv
