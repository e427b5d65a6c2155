# GPU box: the round's evidence in one call -- full GPU test-suite, the default bench line, rocprofv3 kernel stats + PMC
# passes of the bench command (tools/make_profiles.sh), the batch sweep, host-side timings.  Everything under gpurun_out/.
cd $GRAFT_REPO_ROOT; TAG=${1:-r03}; O=gpurun_out/$TAG; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12) > $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo
timeout 900 bash tools/make_profiles.sh $TAG > $O/make_profiles.log 2>&1; tail -3 $O/make_profiles.log
timeout 600 python tools/batch_sweep.py > $O/batch_sweep.txt 2> $O/batch_sweep.err; tail -14 $O/batch_sweep.txt
timeout 200 python tools/host_pieces2.py > $O/host_pieces.txt 2>&1; tail -6 $O/host_pieces.txt
timeout 200 python tools/mode_times.py > $O/mode_times.txt 2>&1; tail -5 $O/mode_times.txt
timeout 600 python tools/batch_sweep_fine.py > $O/batch_sweep_fine.txt 2>&1; tail -19 $O/batch_sweep_fine.txt
timeout 300 python tools/shape_times.py 3000,8,40,2000 5000,8,40,4096 3000,16,128,1500 > $O/strip_times.txt 2>&1; tail -3 $O/strip_times.txt
bash tools/pmc_standalone.sh > $O/pmc_standalone.txt 2>&1; cp gpurun_out/pmc_standalone/*.txt $O/ 2>/dev/null
