"""CPU, world_size 2 over gloo: the data-parallel plumbing of the N>1 path -- identical replicas after the rank-0
broadcast, disjoint utterance shards, and the single all-reduce of the flat gradient buffer giving the full-batch mean
(SURVEY 8e).  No kernel is launched; RCCL replaces gloo on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      AMS_LOG_DIR=os.path.join(tmp, 'log%d' % rank))
    for p in (ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')):
        sys.path.insert(0, p)
    from tests.smoke_step import build_front_dpcl
    trainer, tfds = build_front_dpcl(os.path.join(tmp, 'r%d' % rank), B=2, L=256, W=32, N=8, hop=8, layer_size=8, nb_layers=1, E=4)
    g, model, dist = trainer.graph, trainer.model, trainer.args['dist']
    assert dist.world_size == world and dist.rank == rank
    # 1. replicas identical after the broadcast from rank 0 (each rank initialised with its own RNG stream on purpose)
    if rank == 1:
        for v in g.global_variables():
            v.data.add_(1.0)
    trainer._sync_replicas(dist)
    w = torch.cat([v.detach().reshape(-1) for v in g.global_variables()])
    gathered = [torch.zeros_like(w) for _ in range(world)]
    torch.distributed.all_gather(gathered, w)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    # 2. shards: disjoint, contiguous, covering the global batch
    idx = tfds._batch_indices(tfds.TRAIN, 3)
    # 3. gradient exchange: flat_grad = rank-dependent "per-rank mean gradient" -> averaged full-batch gradient
    opt = model.optimize
    opt.flat_grad.copy_(torch.arange(opt.flat_grad.numel(), dtype=torch.float32) * (rank + 1))
    scale = opt.exchange()
    avg = (opt.flat_grad * scale)
    expect = torch.arange(opt.flat_grad.numel(), dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    # 4. clip uses the norm of the averaged gradient
    opt.clip = 0.5
    opt.flat_grad.copy_(torch.ones_like(opt.flat_grad) * (rank + 1))
    s2 = opt.exchange()
    gn = float(torch.linalg.vector_norm(torch.ones_like(opt.flat_grad) * 1.5))
    # 5. one run folder for the job: the id chosen on rank 0 is the id everywhere (save() writes on rank 0 only, every rank
    #    restores from that folder at the end of Trainer.train)
    ids = [None] * world
    torch.distributed.all_gather_object(ids, model.runID)
    same_id = all(i == ids[0] for i in ids)
    # 6. sparsity term (adapt.py:130-132): p_hat is a batch SUM through a non-linear KL.  Per-rank gradient of
    #    [mean-type term + KL(all-reduced p_hat)] averaged over ranks must equal the single-process gradient on the whole batch.
    #    The product's autograd node (ams_hip/functional.py::SparseKL: all-reduce of p_hat in the forward, x world in the backward)
    #    is driven here with CPU stand-ins for its three HIP kernels (test doubles of ams_abs_colsum_fwd / ams_kl_sparsity_*;
    #    the kernels themselves are checked against the oracle in tests/test_gpu_recipes.py).
    from ams_hip import functional as F, ops as O

    def kl_ref(p_hat, p):
        def logfunc(a, b):
            return a * torch.log(torch.clamp(a, 1e-10, 1.0) / torch.clamp(b, 1e-10, 1.0))
        pt = torch.full((), float(p), dtype=p_hat.dtype)
        return (logfunc(pt, p_hat) + logfunc(1 - pt, 1 - p_hat)).sum()

    def kl_bwd_ref(y2, p_hat, up, p, gscale):
        with torch.enable_grad():
            ph = p_hat.detach().clone().requires_grad_(True)
            kl_ref(ph, p).backward()
        return up * gscale * torch.sign(y2) * ph.grad[None, :]
    O.abs_colsum = lambda y2: y2.abs().sum(0)
    O.kl_sparsity_fwd = lambda p_hat, p: kl_ref(p_hat, p).reshape(1)
    O.kl_sparsity_bwd = kl_bwd_ref
    gen = torch.Generator().manual_seed(5)
    y_all = (torch.rand(2 * world, 6, generator=gen) * 0.1).double()
    ya = y_all.clone().requires_grad_(True)
    ((ya ** 2).sum(1).mean() + kl_ref(ya.abs().sum(0), 0.05)).backward()
    mine = y_all[2 * rank:2 * rank + 2].clone().requires_grad_(True)
    ((mine ** 2).sum(1).mean() + F.sparse_kl(mine, 0.05, dist)).backward()
    # what FlatOptimizer.exchange does to a parameter gradient: sum over ranks x 1/world.  Here the "parameter" is the input
    # shard itself, so compare d/d(shard) x (1/world) with the matching rows of the full-batch gradient: the mean term carries
    # 1/(2*world) vs 1/2 locally (-> x 1/world), the KL term must come out unscaled.
    kl_ok = bool(torch.allclose(mine.grad / world, ya.grad[2 * rank:2 * rank + 2], rtol=1e-10, atol=1e-12))
    # 7. bucketed exchange (AMS_DP_OVERLAP=1; ams_hip/optim.py::bucket_ready): the layers' ranges sent one by one as the backward
    #    pass finishes them + exchange() for what is left == ONE all-reduce of the whole buffer, bit for bit; ranges cover whole
    #    twin-interleaved blocks and never overlap
    opt.clip = 0.0
    opt.overlap, opt._done = True, []
    ref = torch.arange(opt._gbuf.numel(), dtype=torch.float32).mul_(0.37 * (rank + 1)).sin_()
    opt._gbuf.copy_(ref)
    tv = model.trainable_variables
    layer = [v for v in tv if 'BLSTM_0' in v.ams_name]
    dense = [v for v in tv if v.ams_name in ('prediction/W', 'prediction/b')]
    ranges = [opt.range_of(*layer), opt.range_of(*dense)]
    opt.bucket_ready(dense)
    opt.bucket_ready(layer)
    sent = sorted(opt._done)
    opt.exchange()
    bucketed = opt._gbuf.clone()
    opt._gbuf.copy_(ref)
    opt.overlap = False
    opt.exchange()
    n_l, n_d = sum(v.numel() for v in layer), sum(v.numel() for v in dense)
    bucket_ok = (torch.equal(bucketed, opt._gbuf) and sent == sorted(ranges) and ranges[0][1] - ranges[0][0] == n_l
                 and ranges[1][1] - ranges[1][0] == n_d and (ranges[0][1] <= ranges[1][0] or ranges[1][1] <= ranges[0][0]) and not opt._done)
    q.put((rank, same, idx.tolist(), bool(torch.allclose(avg, expect)), abs(s2 - (1.0 / world) * 0.5 / max(gn, 0.5)) < 1e-6,
           same_id, kl_ok, bucket_ok))
    dist.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_data_parallel(tmp_path):
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), 'replicas differ after broadcast'
    assert res[0][2] == [12, 13] and res[1][2] == [14, 15]            # batch 3, world 2, B=2: (3*2+rank)*2 ...
    assert all(r[3] for r in res), 'all-reduced mean gradient wrong'
    assert all(r[4] for r in res), 'global-norm clip must use the averaged gradient'
    assert all(r[5] for r in res), 'run id differs across ranks'
    assert all(r[6] for r in res), 'sparsity (KL of the batch-summed p_hat) gradient is not the single-process gradient'
    assert all(r[7] for r in res), 'bucketed gradient exchange differs from the single all-reduce'
