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
        for i in range(STEPS):
            costs.append(float(model.train(feed, i)))
    torch.cuda.synchronize()
    return costs, {v.ams_name: v.detach().cpu().numpy().copy() for v in model.trainable_variables}, trainer


def _worker(rank, world, port, tmp, B, hip_graph, q):
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


@pytest.mark.parametrize('hip_graph', [False, True])
def test_two_ranks_on_one_gpu_match_single_process(hip_graph):
    import torch.multiprocessing as mp
    B, world = 4, 2
    tmp = tempfile.mkdtemp(prefix='ams_dp_')
    os.environ.setdefault('AMS_LOG_DIR', os.path.join(tmp, 'log'))
    c_one, p_one, _ = _train(B, hip_graph, os.path.join(tmp, 'single'))

    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tmp, B, hip_graph, q)) for r in range(world)]
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
