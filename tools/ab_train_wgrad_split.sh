cd $GRAFT_REPO_ROOT
run() { CMDGEN_OPTIONS=$1 timeout -k 10 120 python tools/bench_train.py --steps 20 --warmup 5 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  loss %.7f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do for o in "wgrad_split=-1" "wgrad_split=1" "wgrad_split=1,wgrad_tile=64" "wgrad_split=0"; do for b in 64 256; do echo -n "[B=$b f32 $o] "; run "$o" "--batch $b"; done; done; done
