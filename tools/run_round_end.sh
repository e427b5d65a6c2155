# GPU box: the round's evidence in one call -- full GPU test-suite, the default bench line, rocprofv3 kernel stats + PMC
# passes of the bench command (tools/make_profiles.sh), cfg 5 (bench line, kernel stats, PMC), sweeps, host-side timings.
# Everything under gpurun_out/<tag>/ ; tools/summarize_profiles.py + a few cp turn it into profiles/<tag>_*.
cd $GRAFT_REPO_ROOT; TAG=${1:-r06}; O=gpurun_out/$TAG; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12) > $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo
timeout 900 bash tools/make_profiles.sh $TAG > $O/make_profiles.log 2>&1; tail -3 $O/make_profiles.log
timeout 900 bash tools/run_cfg5.sh > $O/run_cfg5.log 2>&1; cp gpurun_out/cfg5/bench_cfg5.json $O/ 2>/dev/null; find gpurun_out/cfg5/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cfg5_kernel_stats.csv; tail -5 $O/run_cfg5.log | cut -c1-200
timeout 900 bash tools/pmc_cfg5.sh > $O/pmc_cfg5.log 2>&1; cp gpurun_out/pmc_cfg5/pmc_cfg5.json $O/ 2>/dev/null; for c in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_cfg5/$c.txt $O/pmc_cfg5_$c.txt; done; tail -4 $O/pmc_cfg5.log
timeout 600 python tools/batch_sweep.py > $O/batch_sweep.txt 2> $O/batch_sweep.err; tail -14 $O/batch_sweep.txt
timeout 300 python tools/eager_step_time.py > $O/eager_step_time.txt 2>&1; tail -8 $O/eager_step_time.txt
timeout 300 python tools/overlap_probe.py > $O/overlap_probe.txt 2>&1; tail -6 $O/overlap_probe.txt
timeout 300 bash tools/pmc_issue.sh > $O/pmc_issue.txt 2>&1; tail -4 $O/pmc_issue.txt
timeout 120 python tools/memset_in_graph_probe.py > $O/memset_in_graph.txt 2>&1; tail -3 $O/memset_in_graph.txt
timeout 200 python tools/mode_times.py > $O/mode_times.txt 2>&1; tail -5 $O/mode_times.txt
timeout 300 python tools/shape_times.py 400,64,1500,30 400,64,2100,30 400,64,3000,30 400,64,5000,30 400,96,2500,30 400,64,512,30 1000,64,40,200 > $O/shape_times.txt 2>&1; tail -7 $O/shape_times.txt
bash tools/pmc_standalone.sh > $O/pmc_standalone.txt 2>&1; for c in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_standalone/$c.txt $O/pmc_standalone_$c.txt 2>/dev/null; done
ASG_DTYPE=f64 timeout 300 python tools/shape_times.py 400,64,128,30 400,64,300,30 400,64,512,30 400,64,1024,30 400,64,2048,30 > $O/shape_times_f64.txt 2>&1; tail -5 $O/shape_times_f64.txt
timeout 300 python tools/shape_times.py 400,128,3000,30 200,96,5000,30 >> $O/shape_times.txt 2>&1
timeout 900 python tools/fuzz_routes.py 400 5 > $O/fuzz_400.txt 2>&1; tail -2 $O/fuzz_400.txt
# (needs torch_asg_amd/csrc/variants/libprobes.so: tools/devbuild_generic.sh probes -DASG_DEV_PROBES)
SHAPES="400,64,1500,30 400,64,3000,30 400,64,5000,30 400,32,3000,30 400,16,2100,30 400,128,3000,30" KS="1 2" timeout 900 bash tools/step_grid_time.sh > $O/step_grid.txt 2>&1; grep -c "mb=" $O/step_grid.txt
