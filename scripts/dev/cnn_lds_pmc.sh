# LDS counters of the convolutional step's kernels (levels mode: the step bear_net.train runs), per kernel
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/cnn_lds --output-format csv -- python3 $R/scripts/dev/cnn_order_pmc.py levels > $R/gpurun_out/cnn_lds.log 2>&1 || exit 1
cd $R; python scripts/pmc_table.py cnn gpurun_out/cnn_lds
