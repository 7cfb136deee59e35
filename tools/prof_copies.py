import sys, os
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import torch
import bench_train as bt
dev = torch.device('cuda:0')
cfg, model, tr = bt.build_trainer(64, 'CA', 'fp32', dev)
batches = [bt.synthetic_batch(64, 64 * i, dev) for i in range(2)]
for i in range(4): tr.training_step(batches[i % 2])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(2): tr.training_step(batches[i % 2])
    torch.cuda.synchronize()
evs = prof.events()
for e in evs:
    n = e.name
    if ('emcpy' in n or 'copy_' in n or 'hipMemcpy' in n or 'to' == n or 'aten::to' in n or 'item' in n) and e.device_type.name == 'CPU':
        st = [s for s in (e.stack or []) if 'cmdgen_amd' in s or 'bench_train' in s][:2]
        print(n, [tuple(s) for s in (e.input_shapes or [])][:2], round(e.cpu_time_total, 1), st)
