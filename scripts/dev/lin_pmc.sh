# instruction counters of dm_linear_plan_kernel per developer variant (build_variants/, scripts/dev/lin_variants.py)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in ${VARIANTS:-default linskipbc linskipc}; do
  if [ $n = default ]; then unset BEAR_AMD_LIB; else export BEAR_AMD_LIB=$R/build_variants/libbear_$n.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/lin_pmc_$n --output-format csv -- python3 $R/scripts/dev/lin_variants.py > $R/gpurun_out/lin_pmc_$n.log 2>&1 || exit 1
done
cd $R; for n in ${VARIANTS:-default linskipbc linskipc}; do python scripts/pmc_table.py dm_linear gpurun_out/lin_pmc_$n; done
