"""GPU, the N > 1 path WITH REAL KERNELS on a one-GPU box: two ranks (one process each, AMS_DIST_BACKEND=gloo so that both may use
cuda:0 -- RCCL refuses two ranks on one device) train front_DPCL on their own utterance shard of a global batch of 2 x B/2 and must
land on the weights of a single process training the whole batch of B: rank-0 broadcast, one all-reduce of the flat gradient
buffer per step, 1/world folded into the fused optimizer kernel (ams_hip/optim.py::exchange).  Once eager, once with
--hip_graph (collective + optimizer stay outside the replayed graph)."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(L=1024, W=64, N=16, hop=16, layer_size=16, nb_layers=2, E=8)
STEPS = 4


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train(B, hip_graph, tmp):
    from tests.smoke_step import build_front_dpcl
    trainer, tfds = build_front_dpcl(tmp, B=B, hip_graph=hip_graph, no_summaries=True, **CFG)
    g, model = trainer.graph, trainer.model
    costs = []
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: CFG['L']}
        tfds.initialize(tfds.TRAIN)
        from ams_hip import ops as K
        for i in range(STEPS):
            c = float(model.train(feed, i))
            if K.LSTM_RING != '0' and K.ring_error_pending():
                # a ring launch gave up a bounded wait (two processes time-sharing the CUs): the optimizers of BOTH ranks skipped the
                # update (the word travels with the gradient all-reduce) -- repeat the batch on the per-step kernels, as Trainer.train does
                c = float(model.retrain_last(i))
            costs.append(c)
    torch.cuda.synchronize()
    return costs, {v.ams_name: v.detach().cpu().numpy().copy() for v in model.trainable_variables}, trainer


def _worker(rank, world, port, tmp, B, hip_graph, q, ring=False, overlap=False):
    if overlap:
        os.environ['AMS_DP_OVERLAP'] = '1'           # gradients leave in per-layer buckets behind the side stream (ams_hip/optim.py)
    if ring:
        os.environ['AMS_LSTM_RING'] = '1'            # forced: ams_hip.dist keeps the ring recurrence although the ranks share a GPU
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      AMS_DIST_BACKEND='gloo', AMS_LOG_DIR=os.path.join(tmp, 'log'), HSA_ENABLE_IPC_MODE_LEGACY='0')
    for p in (ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        costs, params, trainer = _train(B // world, hip_graph, os.path.join(tmp, 'r%d' % rank))
        dist = trainer.args['dist']
        assert dist.enabled and dist.world_size == world and torch.distributed.get_backend() == 'gloo'
        q.put((rank, costs, params, None))
        dist.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:                                   # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, None, None, '%s\n%s' % (e, traceback.format_exc())))


@pytest.mark.parametrize('hip_graph,ring,overlap', [(False, False, False), (True, False, False), (False, True, False), (True, True, False),
                                                    (False, False, True), (True, False, True)])
def test_two_ranks_on_one_gpu_match_single_process(hip_graph, ring, overlap):
    """ring: the ranks keep the RING recurrence (AMS_LSTM_RING=1 forced; by default ranks sharing a GPU take the per-step kernels):
    data parallelism + rings + the guard word that rides in the gradient all-reduce, with the trainer's repeat-on-give-up.
    overlap: AMS_DP_OVERLAP=1 -- per-layer gradient buckets all-reduced on a communication stream while the backward pass goes on (with
    gloo under --hip_graph the buckets are skipped and exchange() sends everything: a gloo collective cannot be captured)."""
    import torch.multiprocessing as mp
    B, world = 4, 2
    tmp = tempfile.mkdtemp(prefix='ams_dp_')
    os.environ.setdefault('AMS_LOG_DIR', os.path.join(tmp, 'log'))
    c_one, p_one, _ = _train(B, hip_graph, os.path.join(tmp, 'single'))

    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tmp, B, hip_graph, q, ring, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
    for r in res:
        assert r[3] is None, r[3]
    assert all(p.exitcode == 0 for p in procs)
    (_, c0, p0, _), (_, c1, p1, _) = res
    # replicas stay identical (same averaged gradient, same optimizer state)
    for n in p0:
        assert np.array_equal(p0[n], p1[n]), n
    # the per-step cost is a batch mean of per-utterance terms: mean over the two shards == the single-process cost
    assert np.allclose(0.5 * (np.array(c0) + np.array(c1)), c_one, rtol=2e-5), (c0, c1, c_one)
    # weights after STEPS updates: within 1e-5 of the single-process run (different accumulation order of the batch sum only)
    for n in p_one:
        d = np.abs(p0[n] - p_one[n]).max()
        assert d <= 1e-5 * max(1.0, np.abs(p_one[n]).max()), (n, d)


def test_bench_spawns_the_ranks_it_is_asked_for():
    """`python bench.py --gpus 2` with no launcher in the environment must START two ranks (self re-exec under
    torch.distributed.run) and report n_gpus == 2 -- on the one-GPU box both ranks share cuda:0 over gloo.  A reduced geometry of the
    same step (batch 4 per rank, 4096-sample chunks): this checks the plumbing the driver's scaling run depends on, not a number."""
    import json
    import subprocess
    env = dict(os.environ, AMS_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '4',
                        '--chunk', '4096', '--no-cpu-baseline', '--no-secondary', '--no-native-f32', '--roofline-steps', '0', '--quiet'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['warmup'] == 1
    assert j['config']['global_batch'] == 8 and j['config']['parallelism'] == 'dp2'
    assert j['value'] > 0 and np.isfinite(j['final_cost'])
    # the exchange's own record: backend, world size and one entry per rank as gathered over the process group (RCCL runs add its
    # NCCL_DEBUG=INFO excerpt: `Init COMPLETE ... nranks N`, rings, the all-reduce's algorithm and protocol)
    comm = j['comm']
    assert comm['backend'] == 'gloo' and comm['world_size'] == 2 and sorted(r['rank'] for r in comm['ranks']) == [0, 1]
    assert len(set(r['pid'] for r in comm['ranks'])) == 2 and 'not RCCL' in comm['note']
    assert len(j['rank_ms_per_step']) == 2 and j['allreduce_bytes'] > 0
    # strong scaling: the global batch is fixed and split over the ranks; `value` counts the global batch once per step
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '8',
                        '--scaling', 'strong', '--chunk', '4096', '--no-cpu-baseline', '--no-secondary', '--no-native-f32',
                        '--roofline-steps', '0', '--quiet'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    js = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert js['scaling'] == 'strong' and js['n_gpus'] == 2
    assert js['config']['global_batch'] == 8 and js['config']['batch_per_gpu'] == 4
    assert abs(js['value'] - 8 * js['steps'] / (js['ms_per_step'] * 1e-3 * js['steps'])) <= 1e-3 * js['value']
    # a launcher that started a different number of ranks than --gpus is an error, not a silently wrong n_gpus
    env1 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--batch', '4',
                        '--chunk', '4096', '--no-cpu-baseline', '--no-secondary', '--no-native-f32', '--roofline-steps', '0', '--quiet'],
                       env=env1, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)
