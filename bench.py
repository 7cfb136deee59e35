#!/usr/bin/env python3
"""bench.py - denoising-steps/s of the DiffPhar sampling loop on MI355X.

One "step" of this benchmark = one pass of the hot path over one batch:
``ConditionalDDPM.sample_given_pocket`` on a batch of synthetic CrossDocked-shaped pockets
(BASELINE.json configs[1]: 64 C-alpha pockets, 1000-step DDPM chain, fp32), i.e.
timesteps+1 network evaluations per pocket.  Inputs are resident in HBM before the timed
region; noise is drawn on the device (Philox).  With N GPUs every rank runs its own shard of
pockets (weak scaling, no data-path collective); time = MAX over ranks.

The timed chain keeps the pharmacophore points INSIDE the pocket for all K steps - the geometry a trained model
holds (~500 edges per pocket-evaluation, no dead work).  Random-init weights under the shipped schedule
(noise_precision 1e-5, norm_values [1, 4]) inflate the coordinates by 1/alpha_T = 316, the phar points leave the
pocket and ~89 % of the per-block edge work becomes dead (and is skipped): that chain measures an artefact, so it is
reported beside the headline as `config.drifted_shipped_schedule_chain`, never as `value`.  The headline model is the
same architecture and weight generator with noise_precision 0.1, norm_values [1, 0.25] (1/alpha_T = 3.2, max|x| ~ 17 A).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--timesteps T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work: started WITHOUT a torch.distributed environment and with --gpus N > 1, this process touches no
GPU, starts the N ranks itself (torch.distributed.run, one rank per GPU over RCCL) and relays rank 0's JSON line.

Prints ONE JSON line (rank 0).  Also reported: the algorithmic-FLOP roofline of the dominant kernel measured live
with HIP events on the launch stream, the north-star shape (256 pockets) next to the headline, and the oracle's CPU
rate on the host cores (a baseline, not the target).
"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import cmdgen_amd  # noqa: E402,F401
from cmdgen_amd import hip_backend  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets  # noqa: E402
from cmdgen_amd.collectives import host_barrier, wait_collective  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# dense bf16 MFMA peak of the same guide ("~2.5 PF"): 256 CUs x 4 SIMDs x 1024 FLOP/clk (v_mfma_f32_32x32x16_bf16:
# 32768 FLOP per 32 cycles) x 2.4 GHz.  The split engine EXECUTES six bf16 MFMA FLOPs per algorithmic fp32 FLOP.
PEAK_BF16_MFMA_TFLOPS = 2516.6      # (the fp16 forms take the same cycles: same dense peak)
SPLIT_MFMAS_PER_PRODUCT = 6
# the ceiling of a kernel that runs on the three-piece bf16 split, in ALGORITHMIC fp32 FLOP/s: the pipe's dense peak / 6 (rounds 2-4's ceiling)
PEAK_SPLIT_FP32_EQUIV_TFLOPS = PEAK_BF16_MFMA_TFLOPS / SPLIT_MFMAS_PER_PRODUCT
# ... and of a kernel on the HALF engine (round 5: two fp16 pieces per operand, THREE MFMAs per fp32 product): the same pipe / 3
PEAK_HALF_FP32_EQUIV_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3
PEAK_HBM_TBS = 8.0


def peak_of(mfmas_per_product):
    """ceiling in algorithmic fp32 FLOP/s of a kernel that executes `mfmas_per_product` 16-bit MFMAs per fp32 product (1: the fp32 instruction)"""
    return PEAK_FP32_MFMA_TFLOPS if mfmas_per_product <= 1 else PEAK_BF16_MFMA_TFLOPS / mfmas_per_product


def alg_bytes_per_evaluation(H, L, nodes, edges, n_params):
    """B_alg of SURVEY section 8d: node state read + written once per block, 12 B of index + d0 per edge and block, every parameter once"""
    return L * (nodes * (2 * H + 6) * 4.0 + 12.0 * edges) + 4.0 * n_params


def bounded_config(residue_nf, T):
    """Same architecture and weight generator as the shipped config; noise_precision 0.1 / norm_values [1, 0.25] give 1/alpha_T = 3.2 instead
    of 316, so untrained weights cannot inflate the coordinates (max|x| ~ 17 A): ~500 edges per C-alpha pocket-evaluation for all K steps."""
    return ModelConfig(residue_nf=residue_nf, timesteps=T, noise_precision=0.1, norm_values=(1.0, 0.25))


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--batch', type=int, default=64, help='pockets per GPU')
    p.add_argument('--timesteps', type=int, default=1000, help='DDPM chain length (and T of the model)')
    p.add_argument('--representation', default='CA', choices=['CA', 'full-atom'])
    p.add_argument('--n_phar', type=int, default=15)
    p.add_argument('--no-graph', action='store_true')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=12.0)
    p.add_argument('--north-star-batch', type=int, default=256,
                   help='also time one chain at the north-star shape (pockets on one GPU; 0 = skip; N=1 only)')
    p.add_argument('--gemm', default=None, choices=['split', 'fp32'],
                   help="matrix engine of tiles of >= 32 rows: 'split' (default of the library: fp32-accurate, six bf16 MFMAs "
                        "per fp32 product) or 'fp32' (v_mfma_f32_32x32x2_f32)")
    p.add_argument('--strong', action='store_true',
                   help='strong scaling: BASELINE configs[2] verbatim - ONE batch of --global-batch pockets (default 512) split over the '
                        'N ranks (no data-path collective), instead of --batch pockets per rank')
    p.add_argument('--global-batch', type=int, default=512)
    p.add_argument('--shipped-schedule', action='store_true',
                   help='time the chain of the shipped schedule (noise_precision 1e-5, norm_values [1, 4]) as the headline: with untrained weights '
                        'it drifts out of the pocket and most of its edge work is dead (diagnostic; the default line carries it as a side record)')
    p.add_argument('--no-extra-shapes', action='store_true',
                   help='skip the other records of the default line (trained-geometry chain, full-atom shape, training step)')
    p.add_argument('--rehearse-on-one-gpu', action='store_true',
                   help='N > 1 ranks that all use cuda:0 and rendezvous over gloo: runs the whole multi-rank path (sharded pockets, '
                        'barriers, MAX-over-ranks timing, rank count) on a one-GPU box; the number is NOT a scaling result')
    p.add_argument('--option', action='append', default=[], metavar='KEY=VALUE',
                   help='launch option of every handle of this run (cmdgen_set_option, e.g. --option edge_mt=64 --option node64=0); A/B runs')
    p.add_argument('--force-dist', action='store_true',
                   help='initialise torch.distributed (RCCL) even with ONE rank, so that the fence / all-reduce / MAX path of the multi-GPU run executes on a one-GPU box '
                        '(tests/test_hip_rccl.py); needs RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment')
    p.add_argument('--dry-run-launch', action='store_true',
                   help='launch logic only (CPU, gloo, no sampling): used by tests/test_bench_launch.py')
    return p.parse_args(argv)


# --------------------------------------------------------------------------------------------- launching N ranks
def self_launch(args, argv):
    """--gpus N without a torch.distributed environment: start the N ranks as children of a process that has not
    touched the GPU (never re-exec a process that has) and relay rank 0's line; exit code = the launcher's."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    if r.returncode != 0 or line is None:
        sys.stderr.write(r.stdout[-4000:])
        sys.exit(r.returncode or 1)
    print(line)
    sys.exit(0)


# --------------------------------------------------------------------------------------------- accounting
def kernel_source_sha():
    """Identity of the kernel sources a PMC measurement belongs to (profiles/kernel_traffic.json is stamped with it)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'cmdgen_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'cmdgen_amd', 'csrc', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def coord_flop_per_launch(H, L, coord_edges, nodes, split_proj):
    """EquivariantUpdate.coord_model: 2(H^2+H) per listed edge; with split_proj the launch also carries the next
    block's P|Q projections (4 H^2 per row) in L-1 of the L launches."""
    return 2.0 * (H * H + H) * coord_edges + (nodes * 4.0 * H * H * (L - 1) / L if split_proj else 0.0)


def node_flop_per_launch(H, L, nodes, moving, split_proj=False):
    """Algorithmic FLOP of ONE k_node launch, averaged over the L launches of an evaluation - what the launch needs,
    per row: node_mlp 2*(2H*H) + 2*(H*H) = 6H^2; Q_c 2H^2 on every row, P_c 2H^2 on rows that move only (phar rows:
    pocket rows never receive a coordinate update, egnn_new.py:100-101); P|Q of the next block 4H^2 except in the last
    block (block 0's P|Q belong to k_embed)."""
    per_block = [nodes * 8.0 * H * H + moving * 2.0 * H * H + (nodes * 4.0 * H * H if (l + 1 < L and not split_proj) else 0.0)
                 for l in range(L)]
    return sum(per_block) / L


def executed(c, L):
    """(edges, nodes) summed over evaluations as the kernels EXECUTED them, per block on average: the counters' listed totals minus the dead tiles
    that were skipped (cmdgen_counters.edges_skipped / node_rows_skipped are summed over the L launches of every evaluation)."""
    return c['edges'] - c.get('edges_skipped', 0) / L, c['nodes'] - c.get('node_rows_skipped', 0) / L


def whole_job_flop(H, L, dyn, edges, coord_edges, nodes, moving):
    """F_alg of SURVEY 8d with P_c counted on moving rows only: L*[2(H^2+H)(E+Ec) + 12H^2 N + 2H^2 Nm] + in/out
    embeddings (2*dyn*H per node each)."""
    return L * (2.0 * (H * H + H) * (edges + coord_edges) + 12.0 * H * H * nodes + 2.0 * H * H * moving) + 4.0 * dyn * H * nodes


def cpu_baseline(cfg, sd, batch, rep, n_phar, budget_s):
    """The oracle (port of the reference's eager CPU sequence, validated against the reference for results and wall-clock:
    BASELINE.md section 3, tools/validate_cpu_port.py) timed on the host cores on a bounded sample of the same workload: short
    chains on the same pockets, at the reference's BEST operating point - batch swept over {16, 32, 64, 128} (its N_total^2 edge
    build makes big batches slower per pocket) and torch's intra-op pool sized by a quick calibration (small-tensor eager ops
    get SLOWER with hundreds of threads).  The thread count actually used is reported as `cores`."""
    from oracle import ref_cpu
    p = ref_cpu.to_torch_params(sd)
    c = cfg.as_dict()
    ncpu = os.cpu_count() or 1
    cands = sorted({n for n in (4, 8, 16, 32, 64) if n <= ncpu}) or [ncpu]    # hundreds of threads only get slower (40 s / evaluation at 256)
    batches = [16, 32, 64, 128] if rep == 'CA' else [min(batch, 4), min(batch, 8)]     # (the reference's operating point is its own choice, not the GPU run's batch)
    batches = sorted(set(batches))

    def pocket_of(b):
        pb = make_pockets(b, rep, n_phar=n_phar)
        return pb, {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot), 'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}

    best_n, best_t = cands[0], float('inf')
    sweep = {}
    with torch.no_grad():
        pb, pocket = pocket_of(min(batches, key=lambda b: abs(b - 64)))
        for n in cands:                                       # thread calibration: one 1-step chain (2 evaluations) each
            torch.set_num_threads(n)
            ref_cpu.sample_given_pocket(p, c, pocket, pb.num_nodes_phar, timesteps=1)
            t0 = time.perf_counter()
            ref_cpu.sample_given_pocket(p, c, pocket, pb.num_nodes_phar, timesteps=1)
            dt = time.perf_counter() - t0
            if dt < best_t:
                best_n, best_t = n, dt
            if dt > 1.3 * best_t:
                break
        torch.set_num_threads(best_n)
        K = 4
        best = None
        for b in batches:                                     # batch sweep: an equal share of the budget each
            pb, pocket = pocket_of(b)
            ref_cpu.sample_given_pocket(p, c, pocket, pb.num_nodes_phar, timesteps=1)        # warm
            t0 = time.perf_counter()
            evals = 0
            while True:
                ref_cpu.sample_given_pocket(p, c, pocket, pb.num_nodes_phar, timesteps=K)
                evals += K + 1
                el = time.perf_counter() - t0
                if el >= budget_s / len(batches) or evals >= 400:
                    break
            sweep[b] = b * evals / el
            if best is None or sweep[b] > best[0]:
                best = (sweep[b], b, evals, el)
    v, b, evals, el = best
    return {'value': v, 'unit': 'pocket-steps/s', 'cores': int(best_n), 'kind': 'port',
            'batch_sweep_pocket_steps_per_s': {str(k): round(x, 1) for k, x in sweep.items()}, 'best_batch': b,
            'sample': f'best of batch sizes {batches}: {evals} network evaluations (chains of {K} steps + final decode: phar points inside the '
                      f'pocket, the geometry of the headline chain) of the same model on {b} {rep} pockets, torch {torch.__version__} CPU fp32 with '
                      f'{best_n} of {ncpu} hardware threads (best of {cands}), {el:.1f} s for the best batch'}


# --------------------------------------------------------------------------------------------- the other records of the line
def kernel_table(h, prof, pc, H, L, nl_tot):
    """Per-kernel roofline entries from one profiled eager chain: prof = {kernel: (total ms, launches)} (every launch carried its own
    HIP start / stop events), pc = device counters of that chain.  -> (per_kernel, dominant kernel, launch config, units per launch)."""
    ev = max(pc['evaluations'], 1)
    # units one launch EXECUTES: the last block of a conditional evaluation skips tiles whose new h nobody reads (edges_skipped /
    # node_rows_skipped are summed over launches; every evaluation launches each of the two kernels L times)
    units = {'edge_msg': pc['edges'] / ev - pc.get('edges_skipped', 0) / (ev * L), 'node': pc['nodes'] / ev - pc.get('node_rows_skipped', 0) / (ev * L),
             'edge_coord': pc['edges_phar'] / ev}
    flop_launch = {'edge_msg': 2.0 * (H * H + H) * units['edge_msg'],
                   'node': node_flop_per_launch(H, L, units['node'], nl_tot, False),
                   'edge_coord': coord_flop_per_launch(H, L, units['edge_coord'], pc['nodes'] / ev, False)}
    launch_cfg = {k: h.query(k) for k in ('node_mt', 'edge_mt', 'coord_mt', 'edge_grid', 'coord_grid', 'gemm_split', 'half_engine', 'node16_split', 'node16w', 'node64',
                                           'msg_mfmas_per_product', 'node_mfmas_per_product', 'coord_mfmas_per_product')}
    # node64: the 64-row planes node kernel took the launches (kernels_node64.hip; chosen per layout by tile count)
    mt_of = {'edge_msg': launch_cfg['edge_mt'], 'node': 32 if launch_cfg['node64'] == 32 else 64 if launch_cfg['node64'] else launch_cfg['node_mt'], 'edge_coord': launch_cfg['coord_mt']}
    # the matrix engine each kernel ran on, as the library's launchers resolve it: MFMAs per fp32 product (1 / 6 / 3)
    mpp = {'edge_msg': launch_cfg['msg_mfmas_per_product'], 'node': launch_cfg['node_mfmas_per_product'], 'edge_coord': launch_cfg['coord_mfmas_per_product']}
    per_kernel = {}
    for k, (ms_k, n_k) in prof.items():
        avg = ms_k / max(n_k, 1)
        tf = (flop_launch[k] / (avg * 1e-3) / 1e12) if avg > 0 else 0.0
        peak = peak_of(mpp[k])
        ins = 'v_mfma_f32_16x16x32_' if mt_of[k] == 16 else 'v_mfma_f32_32x32x16_'
        per_kernel[k] = {'total_ms': ms_k, 'launches': n_k, 'avg_launch_ms': avg, 'flop_per_launch': flop_launch[k],
                         'tflops': tf, 'rows_per_tile': mt_of[k], 'mfmas_per_fp32_product': mpp[k],
                         'mfma': (ins + 'f16 x3 per fp32 product (half engine: two fp16 pieces per operand)' if mpp[k] == 3 else
                                  ins + 'bf16 x6 per fp32 product (three bf16 pieces per operand)' if mpp[k] == 6 else
                                  ('v_mfma_f32_16x16x4_f32' if mt_of[k] == 16 else 'v_mfma_f32_32x32x2_f32')),
                         # the ceiling of the pipe the kernel executes on, in algorithmic fp32 FLOP/s: the 16-bit pipe's dense peak / (MFMAs per
                         # fp32 product) for the split engines, the fp32 instruction's otherwise - so no `frac` can exceed 1.  A kernel that moves from
                         # six to three MFMAs per product doubles its ceiling: `frac_of_six_mfma_ceiling` keeps rounds 2-4's yardstick (419.4 TF)
                         'peak': peak, 'frac': tf / peak, 'frac_of_six_mfma_ceiling': tf / PEAK_SPLIT_FP32_EQUIV_TFLOPS,
                         'frac_of_fp32_instruction_peak': tf / PEAK_FP32_MFMA_TFLOPS}
    dom = max(per_kernel, key=lambda k: per_kernel[k]['total_ms'])
    return per_kernel, dom, launch_cfg, units


KERNEL_NAMES = {'edge_msg': 'k_edge_msg (GCL.edge_model + attention + segment sum)',
                'node': 'k_node (GCL.node_model + P_c|Q_c projections + P|Q of the next block)',
                'edge_coord': 'k_edge_coord (EquivariantUpdate.coord_model)'}
PER_KERNEL_KEYS = ('total_ms', 'avg_launch_ms', 'tflops', 'flop_per_launch', 'rows_per_tile', 'mfmas_per_fp32_product', 'mfma', 'peak', 'frac', 'frac_of_six_mfma_ceiling', 'frac_of_fp32_instruction_peak')


def chain_record(cfg, sd, pb, K, dev, stream, use_graph, prof_steps=0, gemm=None, warm_K=None):
    """One warm chain (captures the step graph), one timed chain of K posterior steps on a fresh handle; optionally a short
    eager chain with per-launch events for the dominant kernel's roofline.  -> dict"""
    H, L, dyn = cfg.hidden_nf, cfg.n_layers, cfg.joint_nf + 1
    B, nl_tot = len(pb.size), int(pb.num_nodes_phar.sum())
    h = hip_backend.Handle(cfg.as_dict(), dev.index)
    h.load_state_dict(sd)
    if gemm is not None:
        h.set_gemm_mode(gemm == 'split')
    h.set_layout(pb.num_nodes_phar, pb.size)
    px, poh = torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev)
    with torch.cuda.stream(stream):
        h.sample_chain(px, poh, warm_K or K, noise=None, seed=11, pocket_ids=pb.pocket_index, use_graph=use_graph)     # (captures the step graph: K-independent)
        torch.cuda.synchronize(dev)
        h.reset_counters()
        t0 = time.perf_counter()
        h.sample_chain(px, poh, K, noise=None, seed=12, pocket_ids=pb.pocket_index, use_graph=use_graph)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        c = h.counters()
        st = h.chain_status()
        ev = max(c['evaluations'], 1)
        f_ref = whole_job_flop(H, L, dyn, c['edges'], c['edges_phar'], c['nodes'], ev * nl_tot)      # what the reference computes
        f_alg = whole_job_flop(H, L, dyn, *(executed(c, L)[:1]), c['edges_phar'], executed(c, L)[1], ev * nl_tot)   # what ran
        rec = {'pockets': B, 'posterior_steps': K, 'value': B * (K + 1) / dt, 'unit': 'pocket-steps/s',
               'us_per_denoising_step': 1e6 * dt / (K + 1), 'edges_per_pocket_eval': c['edges'] / ev / B,
               'coord_edges_per_pocket_eval': c['edges_phar'] / ev / B, 'edges_per_s': c['edges'] / dt,
               'whole_step_alg_tflops': f_alg / dt / 1e12,
               # whole job against the ceiling of the engine its tile kernels run on (below, once the engines are known) and against rounds 2-4's
               'whole_job_frac_of_six_mfma_ceiling': f_alg / dt / 1e12 / PEAK_SPLIT_FP32_EQUIV_TFLOPS,
               'whole_job_frac_of_fp32_instruction_peak': f_alg / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
               # SURVEY 8d's byte view: algorithmic bytes (node state once per block, 12 B per edge, the parameters once) / wall / 8 TB/s - not the binding roofline
               'hbm_fraction': alg_bytes_per_evaluation(H, L, c['nodes'] / ev, c['edges'] / ev, sum(int(np.prod(v.shape)) for v in sd.values())) * ev / dt / (PEAK_HBM_TBS * 1e12),
               'reference_work_tflops': f_ref / dt / 1e12,       # the reference's FLOPs of these evaluations per second (not a pipe fraction: dead work is skipped)
               'chain_status': st,
               # share of the reference's per-block edge / node-row work this chain did NOT execute because nobody reads its result
               # (blocks skip tiles beyond L - l hops of a moving node; large in a drifted chain, ~0 where the phar points stay in the pocket)
               'dead_work_skipped': {'edge_visits': c.get('edges_skipped', 0) / max(c['edges'] * L, 1), 'node_row_visits': c.get('node_rows_skipped', 0) / max(c['nodes'] * L, 1)}}
        if prof_steps:
            h.reset_counters()
            h.set_kernel_profiling(True)
            h.sample_chain(px, poh, prof_steps, noise=None, seed=12, pocket_ids=pb.pocket_index, use_graph=False)
            prof = h.kernel_profile()
            h.set_kernel_profiling(False)
            per_kernel, dom, launch_cfg, units = kernel_table(h, prof, h.counters(), H, L, nl_tot)
            rec['launch'] = launch_cfg
            rec['roofline'] = {'bound': 'mfma', 'kernel': KERNEL_NAMES[dom], 'achieved': per_kernel[dom]['tflops'], 'peak': per_kernel[dom]['peak'],
                               'unit': 'TFLOP/s', 'frac': per_kernel[dom]['frac'], 'mfma': per_kernel[dom]['mfma'],
                               'frac_of_six_mfma_ceiling': per_kernel[dom]['frac_of_six_mfma_ceiling'],
                               'frac_of_fp32_instruction_peak': per_kernel[dom]['frac_of_fp32_instruction_peak'],
                               'avg_launch_ms': per_kernel[dom]['avg_launch_ms'], 'flop_per_launch': per_kernel[dom]['flop_per_launch'],
                               'units_per_launch': units[dom], 'timing': f'per-launch HIP events of an eager chain of {prof_steps} steps in the geometry the chain starts from',
                               'per_kernel': {k: {kk: v[kk] for kk in PER_KERNEL_KEYS} for k, v in per_kernel.items()}}
            mpp_dom = per_kernel[dom]['mfmas_per_fp32_product']
        else:
            mpp_dom = h.query('msg_mfmas_per_product')
        rec['whole_job_frac'] = f_alg / dt / 1e12 / peak_of(mpp_dom)       # against the ceiling of the engine the dominant kernel runs on
    h.close()
    return rec


def training_step_record(dev):
    """BASELINE configs[3]'s per-GPU work: one training step (fused loss side, activation-saving forward, backward to parameter
    gradients, adaptive clipping + AdamW) of the C-alpha model on 64 ragged synthetic complexes, fp32 results and bf16 GEMM operands."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench_train
    out = {}
    for gemm in ('fp32', 'bf16'):
        cfg, model, tr = bench_train.build_trainer(64, 'CA', gemm, dev)
        batches = [bench_train.synthetic_batch(64, 50000 + 100 * i, dev) for i in range(4)]
        torch.manual_seed(0)
        dt, losses = bench_train.time_training(tr, batches, 20, 3, dev)
        E, Ec = tr.h.query('train_edges'), tr.h.query('train_coord_edges')
        N, Nl = tr.h.n_phar + tr.h.n_pocket, tr.h.n_phar
        f_fwd = whole_job_flop(cfg.hidden_nf, cfg.n_layers, cfg.joint_nf + 1, E, Ec, N, Nl)
        rec = {'ms_per_step': 1e3 * dt / 20, 'complexes_per_s': 64 * 20 / dt, 'first_loss': losses[0], 'last_loss': losses[-1],
               'graph_of_last_batch': {'nodes': N, 'edges': E, 'coord_edges': Ec},
               # forward = the evaluation's algorithmic FLOP on this graph; the backward pass is twice that (data + weight gradients)
               'alg_gflop_per_step': 3.0 * f_fwd / 1e9, 'whole_step_alg_tflops': 3.0 * f_fwd / (dt / 20) / 1e12,
               'whole_step_frac_of_fp32_instruction_peak': 3.0 * f_fwd / (dt / 20) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
               'whole_step_frac_of_split_ceiling': 3.0 * f_fwd / (dt / 20) / 1e12 / PEAK_SPLIT_FP32_EQUIV_TFLOPS}
        if gemm == 'fp32':
            try:        # the step's dominant kernel and its share of the device time (torch.profiler over 3 more steps)
                from torch.profiler import profile, ProfilerActivity
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    for i in range(3):
                        tr.training_step(batches[i % 4])
                    torch.cuda.synchronize(dev)
                rows = [(e.key, e.device_time_total, e.count) for e in prof.key_averages() if e.device_time_total > 0 and not e.key.startswith(('aten::', 'hip'))]
                tot = sum(r[1] for r in rows)
                rows.sort(key=lambda r: -r[1])
                rec['device_ms_per_step'] = tot / 3e3
                rec['launches_per_step'] = sum(r[2] for r in rows) / 3.0
                rec['top_kernels'] = [{'kernel': k.split('(')[0][:80], 'share_of_device_time': us / tot, 'ms_per_step': us / 3e3, 'launches_per_step': n / 3.0}
                                      for k, us, n in rows[:5]]
            except Exception as e:          # noqa: BLE001  (profiler unavailable: the timing above stands by itself)
                rec['top_kernels'] = f'unavailable: {type(e).__name__}'
        out['fp32_results' if gemm == 'fp32' else 'bf16_gemm_operands'] = rec
        del tr, model
    out['workload'] = ('BASELINE.json configs[3] per GPU: training_step of PharPocketDDPM (lightning_modules.py:245-260) on 64 ragged '
                       'CrossDocked-shaped C-alpha complexes, H=256, L=5, l2 loss, AdamW(amsgrad) + adaptive clipping, pipelined steps')
    return out


def dry_run(args, world, rank):
    """Launch logic without a GPU: N ranks rendezvous over gloo, agree on the rank count, rank 0 prints the line."""
    import torch.distributed as dist
    n_ranks = 1
    B, first = args.batch, rank * args.batch
    if args.strong:                               # the block of the ONE global batch this rank would sample
        from cmdgen_amd.sharding import shard_bounds
        lo, hi = shard_bounds(args.global_batch, world)[rank]
        B, first = hi - lo, lo
    blocks = [[first, first + B]]
    if world > 1:
        dist.init_process_group('gloo')
        ones = torch.ones(1)
        dist.all_reduce(ones)
        n_ranks = int(ones.item())
        mine = torch.tensor([first, first + B], dtype=torch.int64)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        blocks = [g.tolist() for g in got]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'denoising steps/sec', 'value': 0.0, 'unit': 'pocket-steps/s', 'n_gpus': n_ranks,
                          'steps': args.steps, 'warmup': args.warmup, 'dry_run': True, 'scaling': 'strong' if args.strong else 'weak',
                          'pocket_blocks': blocks}))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args, argv)                   # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.dry_run_launch:
        return dry_run(args, world, rank)
    # every Handle below starts with these: CMDGEN_OPTIONS=k=v,... of the environment (what the A/B scripts under tools/ export), then --option k=v
    hip_backend.DEFAULT_OPTIONS.update(hip_backend.parse_options(os.environ.get('CMDGEN_OPTIONS', '')))
    hip_backend.DEFAULT_OPTIONS.update(hip_backend.parse_options(','.join(args.option)))
    dist = None
    one_gpu = args.rehearse_on_one_gpu
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(0 if one_gpu else local_rank)
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank if (dist is not None and not one_gpu) else 0)
    cdev = torch.device('cpu') if one_gpu else dev            # where the collectives' tensors live (gloo: host)
    n_gpus = world
    if dist is not None:                          # the ranks that actually run (RCCL all-reduce of ones)
        ones = torch.ones(1, device=cdev)
        wait_collective(dist.all_reduce(ones, async_op=True))
        n_gpus = int(ones.item())
        assert n_gpus == world, (n_gpus, world)
    if args.gpus != n_gpus and rank == 0:
        print(f'note: --gpus {args.gpus} but {n_gpus} ranks were launched; reporting n_gpus={n_gpus}', file=sys.stderr)

    B, T, rep = args.batch, args.timesteps, args.representation
    first_pocket = rank * B                       # weak scaling: every rank has its own B pockets
    if args.strong:                               # configs[2]: one batch of 512 pockets, contiguous blocks of it per rank
        from cmdgen_amd.sharding import shard_bounds
        lo, hi = shard_bounds(args.global_batch, world)[rank]
        B, first_pocket = hi - lo, lo
        assert B >= 1, 'more ranks than pockets'
    # the headline model: BASELINE's architecture with the bounded schedule, so that the untrained weights cannot inflate the coordinates and the
    # phar points stay inside the pocket for the whole chain (module docstring); --shipped-schedule times the drifting chain instead
    R = 20 if rep == 'CA' else 11
    cfg = ModelConfig(residue_nf=R, timesteps=T) if args.shipped_schedule else bounded_config(R, T)
    sd = make_state_dict(cfg, seed=0)
    h = hip_backend.Handle(cfg.as_dict(), dev.index)
    h.load_state_dict(sd)
    if args.gemm is not None:
        h.set_gemm_mode(args.gemm == 'split')
    pb = make_pockets(B, rep, n_phar=args.n_phar, first_index=first_pocket)     # shard = global pockets [first, first + B)
    h.set_layout(pb.num_nodes_phar, pb.size)
    px, poh = torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev)
    use_graph = not args.no_graph
    stream = torch.cuda.Stream(device=dev)

    def chain(seed, graph=use_graph):
        return h.sample_chain(px, poh, T, noise=None, seed=seed, pocket_ids=pb.pocket_index, use_graph=graph)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            host_barrier(device=dev)          # never a blocking collective on the stream the chain is captured on (collectives.wait_collective)
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            chain(1000 + i)
        fence()
        h.reset_counters()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = chain(i)
        fence()
        elapsed = time.perf_counter() - t0
        cnt = h.counters()
        st = h.chain_status()
    if dist is not None:
        tmax = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        wait_collective(dist.all_reduce(tmax, op=dist.ReduceOp.MAX, async_op=True))
        elapsed = float(tmax.item())
    evals_per_chain = T + 1
    total_pockets = args.global_batch if args.strong else n_gpus * B
    value = total_pockets * evals_per_chain * args.steps / elapsed

    result = None
    if rank == 0:
        H, L, dyn = cfg.hidden_nf, cfg.n_layers, cfg.joint_nf + 1
        nl_tot = int(pb.num_nodes_phar.sum())
        # ---- roofline of the dominant kernel: EVERY launch of the three MFMA kernels during one more chain of the same
        # workload carries its own start / stop HIP events (hipExtLaunchKernelGGL on the launch stream: the dispatch's own
        # begin / end timestamps, what rocprofv3 --kernel-trace reports; eager launches: a graph replay has no per-kernel events).
        # Algorithmic FLOP per launch: edge kernels 2(H^2+H) per listed edge (device counters); node kernel see
        # node_flop_per_launch (what each launch needs, not 14 H^2 on every row).
        with torch.cuda.stream(stream):
            h.reset_counters()
            h.set_kernel_profiling(True)
            h.sample_chain(px, poh, T, noise=None, seed=0, pocket_ids=pb.pocket_index, use_graph=False)
            prof = h.kernel_profile()
            h.set_kernel_profiling(False)
            pc = h.counters()
            # whole-evaluation kernel breakdown on the final chain state
            z = out[0].clone(); z[:, 3:] = 0
            kt = h.profile_evaluation(z.contiguous(), torch.cat([out[1][:, :3], out[1][:, 3:] / cfg.norm_values[1]], 1).contiguous(),
                                      torch.full((B,), 0.5, device=dev))
        # ---- steady-state micro-benchmark: one evaluation at the geometry a TRAINED model holds -
        # phar points uniform in a 5 A ball at the pocket centre (random-init weights let the chain drift away,
        # which roughly halves the edge count).  Graph-replayed, timed with HIP events on the launch stream.
        with torch.cuda.stream(stream):
            rng = np.random.Generator(np.random.PCG64(12345 + rank))
            pm_np = np.repeat(np.arange(B), pb.num_nodes_phar)
            com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
            v = rng.normal(size=(nl_tot, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
            xin = (com[pm_np] + v * 5.0 * np.cbrt(rng.uniform(size=(nl_tot, 1)))).astype(np.float32)
            xh_in = torch.from_numpy(np.concatenate([xin, rng.normal(size=(nl_tot, cfg.phar_nf)).astype(np.float32)], 1)).to(dev)
            xq_in = torch.from_numpy(np.concatenate([pb.x, pb.one_hot / cfg.norm_values[1]], 1).astype(np.float32)).to(dev)
            t_in = torch.full((B,), 0.5, device=dev)
            h.reset_counters()
            ms_eval = h.time_evaluation(xh_in, xq_in, t_in, graph_len=10, replays=20 if rep == 'CA' else 3)
            mc = h.counters()
        dt_micro = ms_eval * 1e-3
        m_ev = max(mc['evaluations'], 1)
        steady = {'us_per_evaluation': 1e3 * ms_eval, 'pocket_evaluations_per_s': B / dt_micro,
                  'edges_per_pocket': mc['edges'] / m_ev / B, 'coord_edges_per_pocket': mc['edges_phar'] / m_ev / B,
                  'edges_per_s': mc['edges'] / m_ev / dt_micro,
                  'alg_tflops': whole_job_flop(H, L, dyn, mc['edges'], mc['edges_phar'], mc['nodes'], m_ev * nl_tot) / m_ev / dt_micro / 1e12,
                  'timing': 'hipGraph of 10 evaluations replayed, HIP events on the launch stream'}
        per_kernel, dom, launch_cfg, units = kernel_table(h, prof, pc, H, L, nl_tot)
        kname = KERNEL_NAMES
        achieved = per_kernel[dom]['tflops']
        avg_ms, launches, flop_per_launch = per_kernel[dom]['avg_launch_ms'], per_kernel[dom]['launches'], per_kernel[dom]['flop_per_launch']
        p_ev = max(pc['evaluations'], 1)
        reference_flop_per_launch = {'edge_msg': 2.0 * (H * H + H) * pc['edges'] / p_ev, 'node': node_flop_per_launch(H, L, pc['nodes'] / p_ev, nl_tot, False),
                                     'edge_coord': per_kernel['edge_coord']['flop_per_launch'] if 'edge_coord' in per_kernel else 0.0}[dom]
        # whole-job algorithmic FLOP for the timed region
        f_ref = whole_job_flop(H, L, dyn, cnt['edges'], cnt['edges_phar'], cnt['nodes'], cnt['evaluations'] * nl_tot)            # what the reference computes
        f_alg = whole_job_flop(H, L, dyn, executed(cnt, L)[0], cnt['edges_phar'], executed(cnt, L)[1], cnt['evaluations'] * nl_tot)   # what ran
        # HBM-side bytes per launch from the PMC passes (tools/collect_traffic.py), only when they were taken on
        # exactly these kernel sources
        traffic, traffic_note = None, 'no profiles/kernel_traffic.json'
        tpath = os.path.join(ROOT, 'profiles', 'kernel_traffic.json')
        sha = kernel_source_sha()
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get('kernel_source_sha') != sha:
                    traffic_note = f"stale: measured on kernel sources {tj.get('kernel_source_sha')}, these are {sha}"
                elif tj.get('workload_batch') != B or tj.get('representation') != rep:
                    traffic_note = 'measured on another workload'
                else:
                    traffic = tj.get('hbm_bytes_per_launch', {}).get(dom)
                    traffic_note = tj.get('source')
            except Exception as e:          # noqa: BLE001
                traffic_note = f'unreadable: {e}'
        result = {
            'metric': 'denoising steps/sec', 'value': value, 'unit': 'pocket-steps/s',
            'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            **({'collectives': f'{dist.get_backend()} (barrier, rank count, MAX of the elapsed time) over {world} rank(s)'} if dist is not None else {}),
            **({'rehearsal': f'{world} ranks sharing cuda:0 over gloo (functional rehearsal of the multi-rank path, not a scaling result)'} if one_gpu and world > 1 else {}),
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': (f'BASELINE.json configs[2]: ONE batch of {args.global_batch} CrossDocked-shaped {rep} pockets sharded over {n_gpus} GPUs '
                             f'({B} on rank 0), no data-path collective; ' if args.strong else '') +
                            f'BASELINE.json configs[{1 if rep == "CA" else 4}]: batch {B} CrossDocked-shaped {rep} pockets per GPU '
                            f'(Np={int(pb.size[0])}, Nl={args.n_phar}), {T}-step DDPM sampling '
                            f'(sample_given_pocket: {evals_per_chain} network evaluations per pocket), '
                            + ('SHIPPED schedule (drifting chain, mostly dead work), ' if args.shipped_schedule else
                               'phar points inside the pocket for the whole chain (bounded schedule: noise_precision 0.1, norm_values [1, 0.25]), ') + f'fp32 '
                            f'({("half matrix engine (two fp16 pieces per operand, three MFMAs per fp32 product): fp32-accurate" if launch_cfg["half_engine"] else "split-bf16 matrix engine on tiles of >= 32 rows: fp32-accurate") if launch_cfg["gemm_split"] else "fp32 MFMA"}); '
                            f'one bench step = one such chain',
                'pockets_per_gpu': B, 'timesteps': T, 'representation': rep,
                'denoiser': f'EGNN hidden_nf={H} n_layers={L} joint_nf={cfg.joint_nf} cutoff={cfg.edge_cutoff}, '
                         f'random-init weights (seed 0), noise_precision={cfg.noise_precision}, norm_values={list(cfg.norm_values)}',
                'hip_graph': use_graph, 'noise': 'on-device Philox4x32-10',
                'us_per_denoising_step': 1e6 * elapsed / (args.steps * evals_per_chain),
                'edges_per_pocket_eval': cnt['edges'] / max(cnt['evaluations'], 1) / B,
                # share of the reference's per-block edge / node-row work the timed chains did NOT execute because nobody reads its result
                'dead_work_skipped': {'edge_visits': cnt.get('edges_skipped', 0) / max(cnt['edges'] * L, 1), 'node_row_visits': cnt.get('node_rows_skipped', 0) / max(cnt['nodes'] * L, 1)},
                'phar_edges_per_pocket_eval': cnt['edges_phar'] / max(cnt['evaluations'], 1) / B,
                'edges_per_s': n_gpus * cnt['edges'] / elapsed,
                'whole_step_alg_tflops': f_alg / elapsed / 1e12,
                'chain_status': st,
                'kernel_ms_one_evaluation': kt,
                'steady_state_evaluation': steady,
                'kernel_source_sha': sha,
                'launch': launch_cfg,
                # north_star: "sampled coords within 1e-4 RMS of reference".  Where it is asserted as an ABSOLUTE bound: the reference's
                # own chain at this config's literal size in the bounded-|x| regime (golden G14: 64 pockets, H=256, L=5, K=T=1000:
                # 7.7e-5 A over the whole batch, types identical; tests/test_hip_parity_r3.py).  With the shipped 1e-5 schedule and
                # UNTRAINED weights a chain inflates coordinates to ~800 A (ulp 6e-5 A): there the tests state the bound relative to |x|.
                'parity': 'coordinate RMS vs the reference <= 1e-4 A absolute at this size in the bounded-|x| regime (G14: 7.7e-5 A, types exact); '
                          'per-step z of the shipped schedule pinned by G15 (K = T = 500) relative to |x|',
            },
            'roofline': {
                'bound': 'mfma', 'kernel': kname[dom],
                'achieved': achieved, 'peak': per_kernel[dom]['peak'], 'unit': 'TFLOP/s',
                'frac': per_kernel[dom]['frac'], 'traffic': traffic, 'traffic_note': traffic_note,
                # which ceiling `peak` / `frac` / `whole_job_frac` are priced on (it follows the dominant kernel's engine; rounds 2-4 used 419.4 TF for every kernel:
                # `frac_of_six_mfma_ceiling` / `whole_job_frac_of_six_mfma_ceiling` keep that yardstick, `frac_of_fp32_instruction_peak` round 2's)
                'yardstick': ('2516.6 TF dense 16-bit MFMA / %d MFMAs per fp32 product' % per_kernel[dom]['mfmas_per_fp32_product']) if per_kernel[dom]['mfmas_per_fp32_product'] > 1
                             else '157.3 TF fp32 MFMA instruction',
                'flop_per_launch': flop_per_launch, 'avg_launch_ms': avg_ms, 'launches_timed': launches,
                'units_per_launch': units[dom],
                # `achieved` is ALGORITHMIC fp32 FLOP/s; `peak` is the ceiling of the pipe the dominant kernel executes on:
                # 157.3 TF for the fp32 instruction, 2516.6 / 6 = 419.4 TF fp32-equivalent for a split-engine kernel (each fp32
                # product = six exact bf16 products on the bf16 pipe, cmdgen_split.h) - so no `frac` can exceed 1.  The
                # fraction of the fp32 INSTRUCTION's peak (what round 2 reported) stays as `frac_of_fp32_instruction_peak`.
                'mfma': per_kernel[dom]['mfma'], 'frac_of_fp32_instruction_peak': per_kernel[dom]['frac_of_fp32_instruction_peak'],
                # round 5: the dominant kernels run on the HALF engine (three fp16 MFMAs per fp32 product): their ceiling is 2516.6 / 3 = 838.9 TF.
                # The same achieved rate against rounds 2-4's ceiling (six MFMAs per product, 419.4 TF):
                'frac_of_six_mfma_ceiling': per_kernel[dom]['frac_of_six_mfma_ceiling'], 'mfmas_per_fp32_product': per_kernel[dom]['mfmas_per_fp32_product'],
                # `flop_per_launch` / `achieved` / `frac` count the work the kernel EXECUTED: since round 3 the last block of a conditional
                # evaluation skips tiles whose output nobody reads.  Priced on the work the REFERENCE does in those launches
                # (every row of every block - what rounds 1-2 reported, and what a recomputation from N, Nl and the launch time gives):
                'frac_on_reference_work': (reference_flop_per_launch / (avg_ms * 1e-3) / 1e12 / per_kernel[dom]['peak']) if avg_ms > 0 else None,
                'reference_flop_per_launch': reference_flop_per_launch,
                'whole_job_frac': f_alg / elapsed / 1e12 / per_kernel[dom]['peak'],               # executed work against the dominant kernel's ceiling
                'whole_job_frac_of_six_mfma_ceiling': f_alg / elapsed / 1e12 / PEAK_SPLIT_FP32_EQUIV_TFLOPS,
                # SURVEY 8d's byte view: algorithmic bytes per evaluation (node state once per block, 12 B per edge, the parameters once) / wall / 8 TB/s
                'hbm_fraction': alg_bytes_per_evaluation(H, L, cnt['nodes'] / max(cnt['evaluations'], 1), cnt['edges'] / max(cnt['evaluations'], 1),
                                                         sum(int(np.prod(v.shape)) for v in sd.values())) * cnt['evaluations'] / elapsed / (PEAK_HBM_TBS * 1e12),
                # algorithmic bytes of ONE launch of the dominant kernel (rows it reads / writes once + index triples + one pass over its weights): `traffic` / this = wasted re-reads
                'traffic_alg_bytes': {'edge_msg': units['edge_msg'] * 12.0 + (pc['nodes'] / p_ev) * 3 * H * 4.0 + 2.0 * H * H * 4,
                                      'node': (pc['nodes'] / p_ev) * 7 * H * 4.0 + 7.0 * H * H * 4,
                                      'edge_coord': units['edge_coord'] * 12.0 + (pc['nodes'] / p_ev) * 2 * H * 4.0 + 2.0 * H * H * 4}[dom],
                'whole_job_reference_work_tflops': f_ref / elapsed / 1e12,                         # the reference's FLOPs of the same evaluations per second
                'whole_job_frac_of_fp32_instruction_peak': f_alg / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                # the north-star also asks for the HBM view: PMC bytes per launch / launch time / 8 TB/s (not the binding roofline)
                'hbm_frac_from_pmc': (traffic / (avg_ms * 1e-3) / (PEAK_HBM_TBS * 1e12)) if traffic else None,
                'per_kernel': {k: {kk: v[kk] for kk in PER_KERNEL_KEYS} for k, v in per_kernel.items()},
            },
        }
        # ---- the other records of the default line (N = 1): every chain below keeps the phar points inside the pocket unless its name says "drifted"
        if n_gpus == 1 and not args.no_extra_shapes and not args.strong and rep == 'CA' and not args.shipped_schedule:
            # north_star: a 256-pocket CrossDocked batch on one GPU, same model, K = T
            ns = chain_record(cfg, sd, make_pockets(args.north_star_batch, 'CA', n_phar=args.n_phar), T, dev, stream, use_graph, prof_steps=16, gemm=args.gemm) \
                if args.north_star_batch and args.north_star_batch != B else None
            if ns is not None:
                ns['workload'] = f'north_star: {args.north_star_batch} C-alpha pockets on one GPU, the headline model, {T}-step chain'
                result['config']['north_star_trained'] = ns
            # BASELINE configs[4]: 256 full-atom pockets (Np=366), 100 strided steps of the T-step model, bounded schedule
            cfg_f = bounded_config(11, T)
            fa = chain_record(cfg_f, make_state_dict(cfg_f, seed=0), make_pockets(256, 'full-atom', n_phar=args.n_phar), T, dev, stream,
                              use_graph, prof_steps=6, gemm=args.gemm, warm_K=16)
            fa['workload'] = f'BASELINE.json configs[4] at its literal size: 256 full-atom pockets (Np=366, Nl=15), all {T} steps of the {T}-step model (bounded schedule), one chain'
            result['config']['fullatom_trained'] = fa
            # the same 64-pocket chain under the SHIPPED schedule: untrained weights drift out of the pocket, most edge work is dead and skipped -
            # an artefact of random weights, reported for completeness only (never the headline, never a vs_baseline)
            cfg_s = ModelConfig(residue_nf=20, timesteps=T)
            dr = chain_record(cfg_s, make_state_dict(cfg_s, seed=0), pb, T, dev, stream, use_graph, prof_steps=0, gemm=args.gemm)
            dr['note'] = ('DRIFTED chain: shipped noise_precision 1e-5 / norm_values [1, 4] with random-init weights inflate |x| to ~800 A; see dead_work_skipped. '
                          'Not a measure of the path on live work.')
            result['config']['drifted_shipped_schedule_chain'] = dr
            # the headline workload on the fp32 matrix instruction (cmdgen_set_gemm_mode(0)) next to the default split-bf16 engine
            if args.gemm is None:
                f32 = chain_record(cfg, sd, pb, T, dev, stream, use_graph, prof_steps=0, gemm='fp32')
                result['config']['fp32_instruction_engine'] = {k: f32[k] for k in ('value', 'unit', 'us_per_denoising_step', 'whole_job_frac_of_fp32_instruction_peak', 'edges_per_pocket_eval')}
            # BASELINE configs[3]'s per-GPU work: the training step
            with torch.cuda.stream(stream):          # (a torch stream like the sampler's records: the legacy default stream is the slower, bracketed path)
                result['config']['training_step'] = training_step_record(dev)
            # the same numbers once more as FLAT scalars (a record that keeps only short values still shows them)
            flat = result['config']
            for tag, r in (('north_star', ns), ('fullatom', fa)):
                if r is None:
                    continue
                flat[tag + '_value'] = round(r['value'], 1)
                flat[tag + '_us_per_step'] = round(r['us_per_denoising_step'], 1)
                flat[tag + '_whole_job_frac'] = round(r['whole_job_frac'], 4)
                flat[tag + '_whole_frac_6mfma'] = round(r['whole_job_frac_of_six_mfma_ceiling'], 4)
                flat[tag + '_dominant_frac'] = round(r['roofline']['frac'], 4)
                flat[tag + '_dom_frac_6mfma'] = round(r['roofline']['frac_of_six_mfma_ceiling'], 4)
                flat[tag + '_dead_edge_frac'] = round(r['dead_work_skipped']['edge_visits'], 4)
                flat[tag + '_hbm_fraction'] = round(r['hbm_fraction'], 5)
            ts = flat['training_step']
            flat['train_ms_fp32'] = round(ts['fp32_results']['ms_per_step'], 3)
            flat['train_ms_bf16_operands'] = round(ts['bf16_gemm_operands']['ms_per_step'], 3)
            # the reference DRIVER's own operating points (generate_phars.py:17-24: n_samples 20, num_nodes_phar 3, --timesteps; configs[0]: one pocket,
            # 8 phar points, 50 steps): small batches, where a step is a chain of 3 + 3 L dependent launches and nothing fills the chip
            cfg_d = bounded_config(20, 500)
            sd_d = make_state_dict(cfg_d, seed=0)
            shapes = {}
            for tag, (b_d, nl_d, k_d) in (('driver_shape', (20, 3, 500)), ('driver_shape_t50', (20, 3, 50)), ('single_pocket', (1, 8, 50))):
                r = chain_record(cfg_d, sd_d, make_pockets(b_d, 'CA', n_phar=nl_d), k_d, dev, stream, use_graph, prof_steps=0, gemm=args.gemm, warm_K=16)
                r['workload'] = (f'{b_d} C-alpha pocket copies x {nl_d} phar points, {k_d} of the T = 500 model\'s steps (generate_phars.py defaults / --timesteps 50; '
                                 f'BASELINE configs[0] on the GPU)')
                r['launches_per_step'] = 3 + 3 * cfg_d.n_layers
                shapes[tag] = r
                flat[tag + '_us_per_step'] = round(r['us_per_denoising_step'], 1)
                flat[tag + '_value'] = round(r['value'], 1)
            flat['launches_per_step'] = 3 + 3 * cfg_d.n_layers
            flat['driver_shapes'] = shapes
        if not args.no_cpu_baseline and n_gpus == 1:      # timed on rank 0 at N=1 only
            result['cpu_baseline'] = cpu_baseline(cfg, sd, B, rep, args.n_phar, args.cpu_seconds)
            cpu_v = result['cpu_baseline']['value']
            result['config']['gpu_over_cpu'] = value / cpu_v           # same model, same geometry (phar points inside the pocket)
            if 'north_star_trained' in result['config']:
                result['config']['north_star_trained']['gpu_over_cpu'] = result['config']['north_star_trained']['value'] / cpu_v
        else:
            result['cpu_baseline'] = None
    if dist is not None:
        host_barrier(device=dev)
        dist.destroy_process_group()
    if rank == 0:
        # the flat scalar records FIRST (right after `workload`): a reader that keeps only the leading short values of `config` still shows them
        c0 = result['config']
        lead = [k for k in c0 if k.startswith(('north_star_', 'fullatom_', 'train_ms_', 'gpu_over_cpu', 'driver_shape_', 'single_pocket_', 'launches_per_step'))
                and not isinstance(c0[k], (dict, list, str))]
        result['config'] = {'workload': c0['workload'], **{k: c0[k] for k in lead}, **{k: v for k, v in c0.items() if k != 'workload' and k not in lead}}
        print(json.dumps(result))


if __name__ == '__main__':
    main()
