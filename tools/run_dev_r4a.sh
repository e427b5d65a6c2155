# GPU box (round 4, session a): full GPU suite with the new parity cases, then the long fuzz at ONE 1e-4 gate
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4a; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest.log; tail -5 $O/pytest.log
(timeout 900 python tools/fuzz_routes.py 500 11 2>&1 | tail -40) > $O/fuzz.log; tail -12 $O/fuzz.log
