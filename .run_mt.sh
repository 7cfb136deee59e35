for cfg in "16 64 16" "32 64 16" "16 32 16" "32 32 32" "64 64 16"; do
  set -- $cfg
  echo "=== B=256 node=$1 edge=$2 coord=$3"; CMDGEN_NODE_MT=$1 CMDGEN_EDGE_MT=$2 CMDGEN_COORD_MT=$3 timeout 300 python bench.py --batch 256 --steps 2 --warmup 1 --timesteps 100 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('value %.0f us/step %.1f roof %.3f whole %.3f' % (j['value'], j['config']['us_per_denoising_step'], j['roofline']['frac'], j['roofline']['whole_job_frac']), {k: round(v,3) for k,v in j['config']['kernel_ms_one_evaluation'].items() if k.endswith('_ms')})
"
done
echo "=== full-atom B=16"; timeout 300 python bench.py --batch 16 --representation full-atom --steps 1 --warmup 1 --timesteps 50 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500
