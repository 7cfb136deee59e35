cd $GRAFT_REPO_ROOT
run() { CMDGEN_OPTIONS=$1 timeout -k 10 120 python tools/bench_train.py --steps 20 --warmup 5 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % (d['ms_per_step']))"; }
for rep in 1 2; do for o in "wgrad_split=0" "wgrad_k128=100000" "wgrad_k128=196608" "wgrad_k128=300000" "wgrad_k128=100000000"; do echo -n "[$o] "; for b in 64 128 256; do echo -n "B=$b "; run "$o" "--batch $b" | tr '\n' ' '; done; echo -n "bf16 B=64 "; run "$o" "--gemm bf16" | tr '\n' ' '; echo; done; done
