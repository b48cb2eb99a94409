# FETCH_SIZE and time of the fused reference-mixing step per developer variant (build_variants/libbear_hip_<name>.so); one counter per pass
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in ${VARIANTS:-default}; do
  if [ $n = default ]; then unset BEAR_AMD_LIB; else export BEAR_AMD_LIB=$R/build_variants/libbear_hip_$n.so; fi
  timeout -k 5 100 python3 $R/scripts/dev/refmix_plan_time.py > $R/gpurun_out/mix_pmc_$n.time.log 2>&1 || exit 1
  timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/mix_pmc_$n --output-format csv -- python3 $R/scripts/dev/refmix_plan_time.py > $R/gpurun_out/mix_pmc_$n.log 2>&1 || exit 1
  echo "$n done"
done
cd $R; for n in ${VARIANTS:-default}; do grep "fused" gpurun_out/mix_pmc_$n.time.log | tail -1; python scripts/pmc_summary.py gpurun_out/mix_pmc_$n | grep refmix; done
