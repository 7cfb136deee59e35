cd $GRAFT_REPO_ROOT
for i in 1 2; do
for o in "wgrad_stream=0" "wgrad_stream=1"; do echo "== $o"; CMDGEN_OPTIONS=$o timeout -k 10 120 python tools/bench_train.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['last_loss'])"; done
done
echo "== b256"; for o in "wgrad_stream=0" "wgrad_stream=1"; do CMDGEN_OPTIONS=$o timeout -k 10 120 python tools/bench_train.py --steps 20 --warmup 5 --batch 256 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['last_loss'])"; done
echo "== bf16"; for o in "wgrad_stream=0" "wgrad_stream=1"; do CMDGEN_OPTIONS=$o timeout -k 10 120 python tools/bench_train.py --steps 30 --warmup 5 --gemm bf16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['last_loss'])"; done
