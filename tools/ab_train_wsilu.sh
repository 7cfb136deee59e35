cd $GRAFT_REPO_ROOT
run() { lib=$1; opt=$2; shift 2; if [ "$lib" = "-" ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$lib; fi
  CMDGEN_OPTIONS=$opt timeout -k 10 120 python tools/bench_train.py --steps 40 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['ms_per_step'])"; }
for rep in 1 2 3; do
echo -n "[f32 prevtrain] "; run build/libcmdgen_hip_prevtrain.so ""
echo -n "[f32 head] "; run build/libcmdgen_hip_head.so ""
echo -n "[f32 cur silu=0] "; run - "wgrad_silu=0"
echo -n "[f32 cur silu=3] "; run - "wgrad_silu=3"
done
for rep in 1 2; do
echo -n "[bf16 head] "; run build/libcmdgen_hip_head.so "" --gemm bf16
echo -n "[bf16 cur silu=0] "; run - "wgrad_silu=0" --gemm bf16
echo -n "[bf16 cur silu=3] "; run - "wgrad_silu=3" --gemm bf16
done
