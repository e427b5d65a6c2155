cd $GRAFT_REPO_ROOT
bash tools/run_round_end.sh r04
bash tools/pmc_standalone.sh > gpurun_out/r04/pmc_standalone.log 2>&1; tail -12 gpurun_out/r04/pmc_standalone.log
