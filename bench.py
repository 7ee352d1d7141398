#!/usr/bin/env python
"""Headline benchmark: mixtures/sec of the front_DPCL training step (2 speakers, 256-filter adaptive front
+ 3xBLSTM(600) deep clustering, batch 64 per GPU) -- BASELINE.json `metric`, SURVEY.md 8(d) cfg3(i).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = forward + backward + gradient all-reduce (N>1) + AMSGrad update on one synthetic batch that is
already resident in HBM.  Rank 0 prints ONE JSON line.  The `roofline` object is measured with HIP events
around every launch of the dominant kernel inside the timed region; `cpu_baseline` times the torch-CPU restatement
of the same step (oracle/torch_step.py, float32, all host threads, batch 64) on rank 0 at N=1 only.
"""
from __future__ import print_function

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0    # same guide: v_mfma_f32_32x32x16_bf16 dense peak
# The products run on the 16-bit matrix pipe as f32 arithmetic (csrc/gemm.hip): "fp16x3" where the caller has a bound of each operand
# (every product of this step, the front conv included since round 5: both f32 operands scaled by a power of two and split EXACTLY into two fp16 terms,
# three fp16 MFMA products per f32 product), "bf16x6" otherwise (three bf16 terms, six products) -- f32-level error either way
# (tests/test_gpu_gemm_f16.py, test_gpu_gemm_x6.py).  The roofline prices the ALGORITHMIC f32 flops (2 M N K) against what that
# pipe can deliver for them: 16-bit MFMA peak / (MFMA products issued per f32 product, flop-weighted over the launches).
X6_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='mixtures per GPU (weak scaling) / in total (strong scaling)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help="weak (headline): --batch mixtures per GPU; strong: --batch mixtures in total, split evenly over the ranks "
                         "(SURVEY 8d: global batch 64)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the cfg3(ii) fine-tuning step timing (N=1 only)')
    ap.add_argument('--no-native-f32', action='store_true', help='skip the second timing with native f32 MFMA products (N=1 only)')
    ap.add_argument('--cpu-batch', type=int, default=64)
    ap.add_argument('--cpu-steps', type=int, default=10)
    ap.add_argument('--roofline-steps', type=int, default=5)
    ap.add_argument('--graph', type=int, default=int(os.environ.get('AMS_BENCH_GRAPH', '1')),
                    help='replay fwd+bwd from a captured hipGraph')
    ap.add_argument('--chunk', type=int, default=20480)
    ap.add_argument('--filters', type=int, default=256)
    ap.add_argument('--quiet', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args()


def build(args, tmp):
    os.environ.setdefault('AMS_LOG_DIR', os.path.join(tmp, 'log'))
    if args.scaling == 'strong':
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if args.batch % world:
            raise SystemExit('--scaling strong: global batch %d is not divisible by %d ranks' % (args.batch, world))
        args.global_batch = args.batch
        args.batch = args.batch // world           # per-rank shard of the fixed global batch
    from ams_hip import testing
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Trainer
    folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre_rank%s' % os.environ.get('RANK', '0')),
                                                   window_size=1024, filters=args.filters, hop_size=256,
                                                   chunk_size=args.chunk, batch_size=args.batch, nb_speakers=2)
    a = dict(params)
    a.update(testing.SEPARATOR_DEFAULTS)
    a.update(model_folder=folder, model_previous=None, batch_size=args.batch, learning_rate=1e-3, optimizer='Adam',
             pretraining=False, layer_size=600, nb_layers=3, embedding_size=40, synthetic_batches=2, synthetic_pool=2,
             no_summaries=True, hip_graph=bool(args.graph))
    trainer = Front_Separator_Trainer(DPCL, 'front_DPCL', **a)
    dist, tfds = trainer.prepare()
    # benches use the SURVEY 8(d) dense init U(+-0.05) (the reference's +-12 range is kept for parity fixtures)
    import torch
    g = trainer.graph
    gen = torch.Generator(device='cpu').manual_seed(9)
    W = g.variables['prediction/W']
    W.data.copy_((torch.rand(W.shape, generator=gen) * 0.1 - 0.05).to(W.device))
    trainer._sync_replicas(dist)
    return trainer, tfds, dist


def comm_record(dist, log_path):
    """Who took part in the gradient exchange, as the process group and the collective library report it: backend, world size, every
    rank's host / process / device (gathered over the process group itself), the library version, and -- RCCL -- rank 0's excerpt of
    NCCL_DEBUG=INFO: the communicator's `Init COMPLETE` line (rank, nranks, device, bus id), the rings / trees it built, and the first
    AllReduce line of the gradient buffer's size with the algorithm and protocol chosen for it."""
    import socket
    import torch
    import torch.distributed as td
    me = {'rank': dist.rank, 'host': socket.gethostname(), 'pid': os.getpid(), 'device': torch.cuda.current_device()}
    try:
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        me['device_name'] = pr.name
        for k in ('uuid', 'pci_bus_id', 'pci_device_id', 'pci_domain_id'):
            if hasattr(pr, k):
                me[k] = str(getattr(pr, k))
    except Exception:
        pass
    ranks = [None] * dist.world_size
    td.all_gather_object(ranks, me)
    rec = {'backend': td.get_backend(), 'world_size': td.get_world_size(), 'ranks': ranks,
           'distinct_devices': len(set((r['host'], r.get('uuid') or r.get('pci_bus_id') or r['device']) for r in ranks))}
    if rec['backend'] == 'nccl':
        try:
            rec['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
    else:
        rec['note'] = 'backend %s: not RCCL (several ranks sharing one GPU on a test box; AMS_DIST_BACKEND)' % rec['backend']
    if dist.rank == 0 and log_path:
        path = log_path.replace('%h', socket.gethostname()).replace('%p', str(os.getpid()))
        try:
            lines = open(path, errors='replace').read().splitlines()
            pick = [ln for ln in lines if 'Init COMPLETE' in ln or 'nranks' in ln and 'Init START' in ln]
            pick += [ln for ln in lines if ' Ring ' in ln or ' Trees ' in ln or 'Connected all' in ln][:6]
            ar = [ln for ln in lines if 'AllReduce' in ln]
            big = [ln for ln in ar if 'count %d' % 0 not in ln]
            pick += big[-2:]
            rec['rccl_debug_excerpt'] = [ln[-300:] for ln in pick[:14]]
            rec['rccl_allreduce_lines'] = len(ar)
        except Exception as e:
            rec['rccl_debug_excerpt'] = 'unavailable (%s: %s)' % (type(e).__name__, e)
    return rec


def _newest_profile(suffix):
    """(parsed JSON, repo-relative path) of the newest profiles/rNN_<letter><suffix> (highest round, then highest letter), or (None, None)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_[a-z]' + suffix)))
    if not cands:
        return None, None
    try:
        return json.load(open(cands[-1])), os.path.relpath(cands[-1], ROOT)
    except Exception:
        return None, None


def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(args):
    """The same front_DPCL step (forward + backward + AMSGrad, float32) as a torch-CPU / oneDNN program on the host cores:
    oracle/torch_step.py, the formulation SURVEY 8(d) / BASELINE.md 3 name as the CPU baseline ("what TF-CPU/Eigen+MKL would also
    reduce to").  kind 'port': the reference's own TF-1.4 CPU path cannot run here.  Same synthetic mixtures, same batch size as
    the GPU step.  The thread count is the FASTEST of {1/8, 1/4, 1/2, all} of the hardware threads, probed upwards on one step each
    until a setting is slower than the best so far (on the 128-core / 256-thread GPU host 32 threads take 1.7 s per step, 128 take
    5.9 s and 256 did not finish a probe in five minutes: the fused LSTM kernel does not scale) -- a baseline
    is only fair at its best setting; then 2 more warm-up steps + >= 10 timed ones, median."""
    import numpy as np
    import torch
    from oracle import step as ostep, torch_step
    from data.dataset import synthetic_mixtures
    B = args.cpu_batch
    ncpu = os.cpu_count() or 1
    rng = np.random.RandomState(1)
    P = ostep.init_params(rng, np.float32, front_W=1024, N=args.filters, D_in=args.filters, layer_size=600, nb_layers=3,
                          E=40, F=args.filters, conv1d_scale=0.05)
    mix, nm, _ = synthetic_mixtures(np.arange(B), 2, args.chunk)
    xm, xn = torch.from_numpy(np.ascontiguousarray(mix, np.float32)), torch.from_numpy(np.ascontiguousarray(nm, np.float32))
    ts = torch_step.FrontDPCLStep(P, 256, 3, 40, lr=1e-3, dtype=torch.float32)
    probe = {}
    torch.set_num_threads(max(1, ncpu // 8))
    ts.step(xm, xn)                                                           # first touch / allocator warm-up, not a probe
    for nt in sorted(set(max(1, ncpu // d) for d in (8, 4, 2, 1))):          # smallest first: a pathological setting comes last
        torch.set_num_threads(nt)
        t0 = time.time()
        ts.step(xm, xn)
        probe[nt] = time.time() - t0
        if probe[nt] > 1.15 * min(probe.values()):
            break                                    # past the optimum: more threads only get slower (256 on the GPU host: minutes)
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    for _ in range(2):
        ts.step(xm, xn)
    steps = args.cpu_steps if probe[best] * args.cpu_steps < 60.0 else max(3, int(60.0 / max(probe[best], 1e-3)))
    times = []
    for _ in range(steps):
        t0 = time.time()
        ts.step(xm, xn)
        times.append(time.time() - t0)
    t = float(np.median(times))
    return {'value': B / t, 'unit': 'mixtures/s', 'cores': int(best), 'kind': 'port', 'cpu_model': _cpu_model(), 'host_threads': ncpu,
            'thread_probe_s_per_step': {str(k): round(v, 2) for k, v in sorted(probe.items())},
            'sample': 'torch-CPU (oneDNN/MKL, fused LSTM kernel) float32 restatement of the same front_DPCL step -- not TensorFlow -- '
                      'batch %d, warm-up + median of %d steps, %.2f s/step on %d threads (fastest of the probed thread counts)'
                      % (B, steps, t, best)}


def cpu_baseline_guarded(args):
    """cpu_baseline() in a child process with a hard time limit: a host-side pathology must not take the GPU line with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--cpu-batch', str(args.cpu_batch), '--cpu-steps',
           str(args.cpu_steps), '--chunk', str(args.chunk), '--filters', str(args.filters)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    except Exception as e:
        return {'value': None, 'unit': 'mixtures/s', 'cores': None, 'kind': 'port', 'sample': 'not measured: %s: %s' % (type(e).__name__, e)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: re-exec this very command line under
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).  Rank 0's single JSON line is the only thing on stdout."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import torch
    from ams_hip import ops
    tmp = tempfile.mkdtemp(prefix='ams_bench_')
    nccl_log = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and os.environ.get('AMS_DIST_BACKEND', 'nccl') == 'nccl':
        # RCCL's own account of the communicator (ranks, devices, rings) and of the algorithm / protocol it picks for the gradient
        # all-reduce goes to a file per rank; rank 0's excerpt is part of the JSON line (`comm`) -- the record that N ranks on N
        # devices took part does not rest on this script's word alone.  (One log line per collective: nothing next to a 2.8 ms step.)
        nccl_log = os.path.join(tmp, 'rccl_rank%s.log' % os.environ.get('RANK', '0'))
        os.environ.setdefault('NCCL_DEBUG', 'INFO')
        os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,COLL,TUNING')
        os.environ.setdefault('NCCL_DEBUG_FILE', nccl_log)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):        # the trainer echoes its config like the reference; stdout carries ONE JSON line
        trainer, tfds, dist = build(args, tmp)
    model, g = trainer.model, trainer.graph
    rank, world = dist.rank, dist.world_size
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks -- the reported n_gpus would not be '
                         'the requested one' % (args.gpus, world))

    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: args.chunk}

        def one_step(i):
            return model.train(feed, i)

        # untimed set-up, independent of --warmup: two eager steps + the hipGraph capture happen in the first three calls
        for i in range(3 if args.graph else 1):
            one_step(0)
        # the captured step measures its two forms (forward products from pre-split operand images or split in the product) over its
        # next 192 replays (about half a second) and keeps the faster one on THIS board (ops.PS_AUTOTUNE): untimed, like the capture itself
        for i in range(400):
            if (getattr(model, '_cg_state', None) or {}).get('tune') is None:
                break
            one_step(0)
        if ops.PS_TUNED and not ops.PS_TUNED.get('presplit', True):
            ops.PRESPLIT = False                         # the eager / stamped passes below measure the form the timed graph runs
        for i in range(args.warmup):
            c = one_step(i)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ops.PROFILE.reset(enabled=not args.graph)
        opt_ = model.optimize
        if world > 1:
            opt_.exchange_events = []                # HIP events around the gradient all-reduce of every timed step
        t0 = time.perf_counter()
        for i in range(args.steps):
            c = one_step(args.warmup + i)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        ops.PROFILE.enabled = False
        exch_ms = None
        if world > 1 and opt_.exchange_events:
            exch_ms = sum(a.elapsed_time(b) for a, b in opt_.exchange_events) / len(opt_.exchange_events)
        opt_.exchange_events = None
        ops.raise_on_ring_errors()                       # a ring launch that gave up a bounded wait would make this number meaningless
        last_cost = float(c)
        if last_cost != last_cost or last_cost in (float('inf'), float('-inf')):
            raise SystemExit('bench.py: the timed steps ended with cost %r -- a diverged step is not a measurement' % last_cost)
        prof_steps = args.steps
        if args.graph and args.roofline_steps:
            # HIP events recorded during capture cannot be read back after a replay (hipErrorInvalidHandle), so the launches of the
            # dominant kernel are bracketed by device-clock STAMPS (ams_stamp, ops._Profile) in a SECOND capture of the same step:
            # same model, same batches, same streams and overlap as the timed graph, plus two one-thread kernels per timed launch.
            # Its replays give the per-launch durations INSIDE the replayed step (each includes ~3 us of kernel boundaries).
            st = model._cg_state
            timed_graph = st['graph']
            st['graph'], st['n'] = None, 2                         # next call captures again
            ops.PROFILE.reset(enabled=True, stamps=True)
            one_step(args.warmup + args.steps)                      # capture (stamps included) + first replay
            ops.PROFILE.enabled = False                             # the stamps stay in the graph; nothing else is recorded
            prof_steps = 1
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.roofline_steps):
                one_step(args.warmup + args.steps + 1 + i)
            torch.cuda.synchronize()
            stamped_ms = (time.perf_counter() - t1) / args.roofline_steps * 1e3
            del timed_graph
        elif args.graph:
            prof_steps = 0
        # Per-launch durations above are taken INSIDE the step, where weight-gradient products share the chip with the ring
        # recurrence and with each other (side stream, residency cap): two products that run side by side each look half as fast.
        # The same launches timed ALONE (same model, same batches, side stream off, no cap) say what the kernel itself reaches.
        alone = None
        stamped_ms = locals().get('stamped_ms', 0.0)
        if prof_steps and args.roofline_steps:
            from ams_hip import functional as F
            was, main_prof = F.OVERLAP.enabled, ops.PROFILE
            F.OVERLAP.enabled = False
            ops.PROFILE = type(main_prof)()
            was_graph = model.args.get('hip_graph')
            model.args['hip_graph'] = False
            try:
                one_step(args.warmup + args.steps)
                ops.PROFILE.reset(enabled=True)
                alone_steps = min(args.roofline_steps, 3)
                for i in range(alone_steps):
                    one_step(args.warmup + args.steps + 1 + i)
                ops.PROFILE.enabled = False
                torch.cuda.synchronize()
                alone = {'steps': alone_steps, 'family': ops.PROFILE.summary(prefix='gemm'),
                         'by_tag': {t: ops.PROFILE.summary(t) for t in ops.PROFILE.tags() if t.startswith('gemm')}}
            finally:
                F.OVERLAP.enabled, ops.PROFILE = was, main_prof
                model.args['hip_graph'] = was_graph

    # per-rank step times and the exposed (un-overlapped: the all-reduce sits between the replayed graph and the optimizer kernel on one
    # stream) all-reduce time, gathered BEFORE the max over ranks replaces the local figure
    per_rank = torch.zeros(world, dtype=torch.float64, device='cuda')
    per_rank[rank] = elapsed / args.steps * 1e3
    dist.all_reduce_sum(per_rank)
    exch = torch.tensor([exch_ms if exch_ms is not None else 0.0], dtype=torch.float64, device='cuda')
    dist.all_reduce_max(exch)
    el = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    dist.all_reduce_max(el)
    elapsed = float(el.item())
    B, L, N, T, H, E, S = args.batch, args.chunk, args.filters, -(-args.chunk // 256), 300, 40, 2
    value = world * B * args.steps / elapsed

    # ---- roofline of the dominant kernel: the fp32 MFMA GEMM family (dense 600->F*E, its two gradients, LSTM projections)
    prof = ops.PROFILE.summary(prefix='gemm')
    from ams_hip._lib import load as _load
    x6 = bool(_load().ams_gemm_get_arith())
    kname = 'gemm_x6_kernel' if x6 else 'gemm_f32_kernel'
    # MFMA products issued per f32 product: 3 (fp16x3, profile tag gemm16<..>), 6 (bf16x6), 1 (native f32 MFMA)
    issue = lambda tag: (3.0 if tag.startswith('gemm16') else 6.0) if x6 else 1.0     # noqa: E731
    fl16 = sum(ops.PROFILE.summary(t)['flops'] for t in ops.PROFILE.tags() if t.startswith('gemm16'))
    issued = sum(issue(t) * ops.PROFILE.summary(t)['flops'] for t in ops.PROFILE.tags() if t.startswith('gemm'))
    factor = issued / prof['flops'] if prof['flops'] else 1.0
    peak = (MFMA_BF16_PEAK_TFLOPS if x6 else MFMA_F32_PEAK_TFLOPS) / factor
    roof = None
    tj = None
    if prof['launches']:
        avg_ms = prof['ms'] / prof['launches']
        flops_per_launch = prof['flops'] / prof['launches']
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        if not x6:
            kdesc = 'gemm_f32_kernel (v_mfma_f32_32x32x2_f32)'
        elif fl16 > 0.5 * prof['flops']:
            kdesc = ('gemm_x6_kernel<.., F16=true> + gemm_ps_kernel (fp16x3: 3 x v_mfma_f32_32x32x16_f16 per f32 product -- both operands scaled '
                     'by a power of two from a per-tensor bound and split exactly into two fp16 terms, f32 accumulate; the forward products '
                     'from operand images cut once per step, csrc/gemm_ps.hip: %.1f %% of the family\'s flops; fp16x3 in all %.1f %%, the rest '
                     'bf16x6)' % (100.0 * sum(ops.PROFILE.summary(t)['flops'] for t in ops.PROFILE.tags() if t.startswith('gemm16ps')) / prof['flops'],
                                  100.0 * fl16 / prof['flops']))
        else:
            kdesc = kname + ' (6 x v_mfma_f32_32x32x16_bf16 per f32 product: exact 3-way bf16 operand split, f32 accumulate)'
        roof = {'bound': 'mfma', 'kernel': kdesc, 'achieved': round(achieved, 2),
                'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                'traffic': None, 'launches_per_step': prof['launches'] / prof_steps,
                'avg_launch_ms': round(avg_ms, 4), 'share_of_step': round(prof['ms'] / prof_steps / (elapsed / args.steps * 1e3), 3),
                'measured': ('device-clock stamps (ams_stamp) around each launch INSIDE a replayed hipGraph of the same step (a second '
                             'capture with two one-thread stamp kernels per timed launch: %.3f ms per replay against %.3f ms for the '
                             'timed graph; every duration includes ~3 us of kernel boundaries)' % (stamped_ms, elapsed / args.steps * 1e3))
                if args.graph else 'HIP events around each launch inside the timed region',
                'by_variant': {}}
        if x6:
            roof['achieved_unit_note'] = 'f32-equivalent TFLOP/s = algorithmic 2*M*N*K per launch / duration'
            roof['peak_note'] = ('16-bit MFMA dense peak %.0f TFLOP/s / %.2f MFMA products issued per f32 product (3 for fp16x3 launches, 6 for '
                                 'bf16x6, flop-weighted); the same launches reach %.2f of the NATIVE f32 MFMA peak (%.1f TFLOP/s), which this '
                                 'arithmetic is not bound by' % (MFMA_BF16_PEAK_TFLOPS, factor, achieved / MFMA_F32_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS))
            roof['mfma_16bit_flops_issued_TFLOP/s'] = round(factor * achieved, 1)
        # HBM traffic of the same kernels: rocprofv3 PMC passes cannot run inside this process, so the per-launch figure is CITED
        # from the newest committed summary of `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this very workload
        # (tools/pmc_traffic.sh -> tools/pmc_summary.py; gfx950 x2 read correction applied there), tagged with the commit the
        # counters were collected at.  null for any other shape or when no summary is committed.
        tj, tfile = _newest_profile('_hbm_traffic.json')
        if tj is not None and (B, L, N) == (64, 20480, 256):
            gk = [v for k, v in tj.items() if (k.startswith(kname) or k.startswith('gemm_ps_kernel')) and isinstance(v, dict)]
            calls = sum(v['calls'] for v in gk)
            if calls:
                mb = sum(v['calls'] * (v['read_MB_per_launch'] + v['write_MB_per_launch']) for v in gk) / calls
                roof['traffic'] = round(mb * 1e6)
                roof['traffic_unit'] = 'bytes per launch (memory-side L2 requests)'
                roof['traffic_source'] = {'file': tfile, 'counted_at_commit': (tj.get('_meta') or {}).get('commit', '1a73b9d')}
                roof['algorithmic_bytes_per_launch'] = round(prof['bytes'] / prof['launches'])
        # the same kernels inside the REPLAYED (timed) hipGraph: per-variant averages of a rocprofv3 --kernel-trace pass over
        # `bench.py --graph 1` (tools/prof_step.sh), cited by file
        rj, rfile = _newest_profile('_replay_kernels.json')
        if rj is not None and (B, L, N) == (64, 20480, 256):
            roof['replayed_region'] = {'file': rfile, 'counted_at_commit': (rj.get('_meta') or {}).get('commit'),
                                       'by_kernel': {k: v for k, v in rj.items() if k != '_meta'}}
        if alone is not None and alone['family']['launches']:
            fa = alone['family']
            ach = fa['flops'] / (fa['ms'] * 1e-3) / 1e12
            roof['standalone'] = {
                'achieved': round(ach, 2), 'frac': round(ach / peak, 4), 'unit': 'TFLOP/s',   # same arithmetic mix, hence the same peak
                'measured': 'HIP events around each launch, %d eager steps with the side stream off (no co-resident kernel, no residency '
                            'cap): the kernel alone at the step\'s own shapes' % alone['steps'],
                'by_variant': {kname + t[4:]: {'avg_launch_us': round(v['ms'] / v['launches'] * 1e3, 2),
                                                            'TFLOP/s': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)}
                               for t, v in alone['by_tag'].items() if v['launches']},
                'ceiling_note': ('the MFMA-only instruction stream of this kernel (split, LDS and fetch compiled out) runs the 4096^3 product in '
                                 '379 us = 362 TFLOP/s f32-equivalent at 2400 MHz / 1050 W; the whole kernel draws 1385-1393 W, i.e. the board '
                                 'power limit, at 2135 MHz (tools/probes/clock_probe.sh, DESIGN.md 4)') if x6 else
                                ('tools/probes/mfma_peak.hip on the same boxes: dependent-free v_mfma_f32_32x32x2_f32 streams reach 152-156 TFLOP/s '
                                 'for ~100 us and settle at ~125 TFLOP/s when sustained (power management), DESIGN.md 4')}
        for tag in ops.PROFILE.tags():
            if not tag.startswith('gemm'):
                continue
            pv = ops.PROFILE.summary(tag)
            if pv['launches']:
                roof['by_variant'][kname + tag[4:]] = {
                    'launches_per_step': pv['launches'] / prof_steps, 'avg_launch_us': round(pv['ms'] / pv['launches'] * 1e3, 2),
                    'TFLOP/s': round(pv['flops'] / (pv['ms'] * 1e-3) / 1e12, 2)}

    # ---- second roofline entry: the HBM-bound loss kernels (fused l2norm + DPCL Gram pass, its backward)
    roof_hbm = None
    pd = ops.PROFILE.summary(label='dpcl')
    if pd['launches']:
        ach = pd['bytes'] / (pd['ms'] * 1e-3) / 1e9
        roof_hbm = {'bound': 'hbm', 'kernel': 'dpcl_gram_u_kernel + dpcl_bwd_u_kernel', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': None, 'by_kernel': {},
                    'measured': 'HIP events around each launch, same steps as `roofline`'}
        for tag in ('dpcl_gram_u', 'dpcl_bwd_u'):
            pv = ops.PROFILE.summary(tag)
            if pv['launches']:
                roof_hbm['by_kernel'][tag + '_kernel'] = {
                    'avg_launch_us': round(pv['ms'] / pv['launches'] * 1e3, 2), 'GB/s': round(pv['bytes'] / (pv['ms'] * 1e-3) / 1e9, 1),
                    'algorithmic_bytes_per_launch': round(pv['bytes'] / pv['launches'])}
        if tj is not None and (B, L, N) == (64, 20480, 256):
            dk = [v for k, v in tj.items() if k.startswith('dpcl_gram_u') or k.startswith('dpcl_bwd_u')]
            if dk:
                roof_hbm['traffic'] = round(sum(v['read_MB_per_launch'] + v['write_MB_per_launch'] for v in dk) / len(dk) * 1e6)
                roof_hbm['traffic_source'] = roof.get('traffic_source') if roof else None

    # ---- the two kernels north_star names a target for (same HIP-event records)
    targets = {}
    pf = ops.PROFILE.summary(label='front_conv')
    if pf['launches']:
        t = pf['ms'] * 1e-3
        conv16 = any(r[5] == 'front_conv' and r[4].startswith('gemm16') for r in ops.PROFILE.records)
        targets['front_conv'] = {
            'kernel': kname + '<A_FRAMES> (stream-K inside the launch; max |y| folded in the launch)', 'avg_launch_us': round(pf['ms'] / pf['launches'] * 1e3, 2),
            'TFLOP/s': round(pf['flops'] / t / 1e12, 2),
            'mfma_frac': round(pf['flops'] / t / 1e12 / ((MFMA_BF16_PEAK_TFLOPS / (3.0 if conv16 else 6.0)) if x6 else MFMA_F32_PEAK_TFLOPS), 4),
            'arithmetic': (('fp16x3 (3 MFMA products per f32 product; bounds from the staging launch and the frozen-filter cache)' if conv16 else
                            'bf16x6 (6 MFMA products per f32 product)') if x6 else 'native f32 MFMA'),
            'vs_native_f32_mfma_peak': round(pf['flops'] / t / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
            'algorithmic_GB/s': round(pf['bytes'] / t / 1e9, 1), 'hbm_frac': round(pf['bytes'] / t / 8e12, 4),
            'note': 'strided analysis conv = dense contraction, AI ~ 250 flop/B: MFMA-bound, not HBM-bound (DESIGN.md 4)'}
    pi = ops.PROFILE.summary(label='blstm_input_gemm')
    if pi['launches']:
        t = pi['ms'] * 1e-3
        targets['blstm_input_gemm'] = {
            'kernel': ('gemm_ps_kernel (pre-split fp16 operand images, LDS-DMA main loop)' if any(t.startswith('gemm16ps') for t in ops.PROFILE.tags())
                       else kname + '<A_ROW,B_ROW>') + ', both directions in one [B*T, D] x [D, 8H] product',
            'avg_launch_us': round(pi['ms'] / pi['launches'] * 1e3, 2), 'TFLOP/s': round(pi['flops'] / t / 1e12, 2),
            'mfma_frac': round(pi['flops'] / t / 1e12 / ((MFMA_BF16_PEAK_TFLOPS / (3.0 if fl16 > 0 else 6.0)) if x6 else MFMA_F32_PEAK_TFLOPS), 4),
            'arithmetic': ('fp16x3 (3 MFMA products per f32 product)' if fl16 > 0 else 'bf16x6') if x6 else 'native f32 MFMA',
            'vs_native_f32_mfma_peak': round(pi['flops'] / t / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}

    out = {
        'metric': 'mixtures/sec training throughput (2-spk, 256-filter adapt+BLSTM-DPCL)',
        'value': round(value, 2), 'unit': 'mixtures/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': ('f32 (storage, accumulation and results f32; dense products issued on the 16-bit matrix pipe as fp16x3 / bf16x6 emulation of '
                  'the f32 product -- `value_native_f32` is the same step on v_mfma_f32_32x32x2_f32)') if x6 else 'f32',
        'data': 'synthetic',
        'arith': ('f32 throughout; dense products on the 16-bit matrix pipe as f32 arithmetic -- fp16x3 (operands scaled by a power of two and split exactly in two fp16 terms, 3 products) where operand bounds are at hand, else bf16x6: exact 3-way bf16 split of both f32 operands, 6 of 9 partial '
                  'products (dropped: <= 2^-26 |a.b|), f32 accumulation -- error vs float64 at or below the native f32 MFMA kernel\'s '
                  '(tests/test_gpu_gemm_x6.py); `secondary.native_f32_mfma` is the same step with v_mfma_f32_32x32x2_f32 products')
        if x6 else 'f32 throughout, products on v_mfma_f32_32x32x2_f32',
        'config': {'workload': 'front_DPCL training step (SURVEY 8d cfg3(i)): frozen adaptive front W=1024 hop=256 N=%d -> '
                               '3xBLSTM(600) -> dense 600->%d -> l2norm -> DPCL loss, fwd+bwd+AMSGrad' % (N, N * E),
                   'batch_per_gpu': B, 'global_batch': B * world, 'nb_speakers': S, 'chunk_size': L, 'frames': T,
                   'parallelism': 'dp%d' % world, 'hip_graph': bool(args.graph)},
        'roofline': roof, 'roofline_hbm': roof_hbm, 'north_star_targets': targets, 'final_cost': last_cost,
    }
    if ops.PS_TUNED:
        out['forward_products'] = dict(ops.PS_TUNED, note='the captured step was replayed both ways before the timed region and kept the faster form on this board: pre-split operand images (csrc/gemm_ps.hip) raise the matrix-pipe duty, and some boards then settle at a lower clock for the whole step (DESIGN.md 4.1b)')
    if world > 1:
        out['rank_ms_per_step'] = [round(float(v), 3) for v in per_rank.cpu().numpy()]
        out['exposed_allreduce_ms'] = round(float(exch.item()), 4)
        out['allreduce_bytes'] = int(model.optimize._gbuf.numel() * 4)
        out['comm'] = comm_record(dist, nccl_log if os.environ.get('NCCL_DEBUG_FILE') == nccl_log else os.environ.get('NCCL_DEBUG_FILE'))
    if rank == 0:
        if world == 1 and not args.no_secondary and (B, L, N) == (64, 20480, 256):
            # BASELINE configs[2] names the fine-tuning flavour of the same model; SURVEY 8(d) makes cfg3(i) the headline and
            # cfg3(ii) a second number: reported here for completeness, never as `value`
            try:
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                import contextlib
                import bench_configs
                with contextlib.redirect_stdout(sys.stderr):
                    r = bench_configs.wl_front_dpcl_finetuning(10, 4, B=B, graph=bool(args.graph))
                out['secondary'] = {'front_DPCL_finetuning_step': {'mixtures_per_s': round(r['mixtures_per_s'], 1),
                                                                   'ms_per_step': round(r['ms_per_step'], 3),
                                                                   'kmeans_seeding': bench_configs.SEEDING,
                                                                   'workload': r['workload']}}
            except Exception as e:                       # the headline line must still be printed
                out['secondary'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if world == 1 and x6 and not args.no_native_f32:
            # the same timed region with the native f32 MFMA products (AMS_GEMM_X6=0), in a fresh process (the hipGraph is captured per arithmetic)
            import subprocess
            try:
                env = dict(os.environ, AMS_GEMM_X6='0')
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(args.warmup),
                                    '--batch', str(args.batch), '--chunk', str(args.chunk), '--filters', str(args.filters),
                                    '--graph', str(int(bool(args.graph))), '--roofline-steps', '0', '--no-secondary', '--no-cpu-baseline',
                                    '--quiet'], env=env, capture_output=True, text=True, timeout=600)
                j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
                out.setdefault('secondary', {})['native_f32_mfma'] = {'mixtures_per_s': j['value'], 'ms_per_step': j['ms_per_step'],
                                                                      'note': 'same step, same run, products on v_mfma_f32_32x32x2_f32'}
                out['value_native_f32'] = j['value']            # the headline metric under the reference's own product arithmetic
            except Exception as e:
                out.setdefault('secondary', {})['native_f32_mfma'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_guarded(args)
        print(json.dumps(out))


if __name__ == '__main__':
    main()
