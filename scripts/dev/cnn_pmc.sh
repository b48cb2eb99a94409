# SQ counters of the cnn backward kernels (both tile forms) at 1e7 contexts
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for form in 2 1; do
  export BEAR_CNN_BACKWARD=$form
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU -d $R/gpurun_out/cnn_pmc_$form --output-format csv -- python3 $R/scripts/dev/cnn_ab.py 1e7 /tmp/o.pt > $R/gpurun_out/cnn_pmc_$form.log 2>&1 || exit 1
done
cd $R; for form in 2 1; do python scripts/pmc_table.py cnn_backward gpurun_out/cnn_pmc_$form; done
