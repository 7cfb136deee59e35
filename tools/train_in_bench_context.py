"""The training step timed as bench.py times it (other handles and a torch stream alive in the process), on the legacy default stream and on a
torch stream: the hardware-queue sharing that cost 2.7 -> 4.5 ms per step in round 6 shows only here, not in a fresh bench_train.py process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import cmdgen_amd  # noqa: F401
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
import bench_train
dev = torch.device('cuda', 0)
stream = torch.cuda.Stream(device=dev)
cfg = ModelConfig(residue_nf=20, timesteps=1000, noise_precision=0.1, norm_values=(1.0, 0.25))
h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=0))
pb = make_pockets(64, 'CA'); h.set_layout(pb.num_nodes_phar, pb.size)
with torch.cuda.stream(stream):
    h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), 50, seed=1, pocket_ids=pb.pocket_index)
    torch.cuda.synchronize()
for where in ('legacy default stream', 'torch stream'):
    _, model, tr = bench_train.build_trainer(64, 'CA', 'fp32', dev)
    batches = [bench_train.synthetic_batch(64, 50000 + 100 * i, dev) for i in range(4)]
    if where == 'torch stream':
        with torch.cuda.stream(stream):
            dt, _ = bench_train.time_training(tr, batches, 20, 3, dev)
    else:
        dt, _ = bench_train.time_training(tr, batches, 20, 3, dev)
    print('%-22s %.3f ms per step' % (where, 1e3 * dt / 20), flush=True)
    del tr, model
