# GPU box: cfg 5 step with the one-launch forward (default) and with T-1 launches (ASG_NO_PERSIST=1)
cd $GRAFT_REPO_ROOT
for e in 0 1; do echo "ASG_NO_PERSIST=$e"; ASG_NO_PERSIST=$e timeout 100 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.1f  utt/s %.2f  fwd kernel us/frame %.1f  frac %.3f  loss %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'] * 1e3, d['roofline']['frac'], d.get('loss')))"; done
