# instruction / LDS counters of the fused linear step, plain against paired lists (profiles/r04*_linear_head_counters.txt)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d /tmp/lin_pmc_a --output-format csv -- python3 $R/scripts/dev/lin_pmc_r04.py > $R/gpurun_out/lin_pmc_r04_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d /tmp/lin_pmc_b --output-format csv -- python3 $R/scripts/dev/lin_pmc_r04.py > $R/gpurun_out/lin_pmc_r04_b.log 2>&1
cd $R; python3 scripts/pmc_table.py dm_linear /tmp/lin_pmc_a /tmp/lin_pmc_b > gpurun_out/lin_pmc_r04.txt 2>&1; tail -3 gpurun_out/lin_pmc_r04_a.log; cat gpurun_out/lin_pmc_r04.txt
