"""Probe for the RCCL-watchdog / stream-capture hazard (cmdgen_amd/collectives.py): one rank, nccl backend; N times
[a collective -> a little eager GPU work and a sleep that sweeps the watchdog's poll period -> a capture on the chain's stream].
    python tools/rccl_capture_probe.py default-blocking | stream-blocking | async [N] [graph_len]
default-blocking: dist.all_reduce on torch's DEFAULT stream (bench.py's MAX-of-elapsed-time reduction until round 6);
stream-blocking:  dist.barrier inside `with torch.cuda.stream(stream)` (bench.py's fence until round 6);
async:            the fixed form (async_op=True + wait).
A hit aborts the process: "operation not permitted on an event last recorded in a capturing stream" from the watchdog thread."""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from cmdgen_amd import hip_backend
from cmdgen_amd.collectives import wait_collective
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets

mode = sys.argv[1] if len(sys.argv) > 1 else 'async'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
G = int(sys.argv[3]) if len(sys.argv) > 3 else 200
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
cfg = ModelConfig(residue_nf=20, timesteps=500)
h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=0))
pb = make_pockets(8, 'CA'); h.set_layout(pb.num_nodes_phar, pb.size)
nl = int(pb.num_nodes_phar.sum())
xh = torch.randn(nl, 3 + cfg.phar_nf, device=dev)
xq = torch.randn(len(pb.mask), 3 + cfg.residue_nf, device=dev)
t = torch.full((8,), 0.5, device=dev)
stream = torch.cuda.Stream(device=dev)
tens = torch.ones(1, device=dev)
t_cap = []
for i in range(N):
    torch.cuda.synchronize(dev)
    if mode == 'default-blocking':
        dist.all_reduce(tens, op=dist.ReduceOp.MAX)
    elif mode == 'stream-blocking':
        with torch.cuda.stream(stream):
            dist.barrier()
    else:
        with torch.cuda.stream(stream):
            wait_collective(dist.barrier(async_op=True))
    torch.cuda.synchronize(dev)
    time.sleep(0.0017 * (i % 60))                    # 0 .. 100 ms: wherever the watchdog's next poll falls, some iteration's capture covers it
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        h.time_evaluation(xh, xq, t, graph_len=G, replays=1)      # BeginCapture .. EndCapture over G evaluations (18 launches each)
        t_cap.append(time.perf_counter() - t0)
print(f'{mode}: {N} captures of {G} evaluations behind {N} collectives, no abort; a call takes {1e3 * sum(t_cap) / len(t_cap):.1f} ms')
dist.destroy_process_group()
