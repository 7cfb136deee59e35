timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=line 2>&1 | tail -3
python tools_ablate.py 256 2>&1 | grep -v amdgpu | head -3
QUICK=1 python tools_ablate.py 16 full-atom 2>&1 | grep -v amdgpu | head -3
for b in 64 256; do timeout 300 python bench.py --batch $b --steps 2 --warmup 1 --timesteps 200 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('B', j['config']['pockets_per_gpu'], 'value %.0f us/step %.1f roof %.3f whole %.3f' % (j['value'], j['config']['us_per_denoising_step'], j['roofline']['frac'], j['roofline']['whole_job_frac']), {k: round(v,3) for k,v in j['config']['kernel_ms_one_evaluation'].items() if k.endswith('_ms')})
"; done
