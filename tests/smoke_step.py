"""One tiny front_DPCL training step on cuda:0, checked against the CPU oracle (used by
__graft_entry__.smoke() and tests/test_gpu_step.py)."""
import os
import tempfile


def build_front_dpcl(tmp, B=3, L=1024, W=64, N=16, hop=16, layer_size=16, nb_layers=2, E=8, S=2, optimizer='Adam',
                     lr=1e-3, **extra):
    os.environ.setdefault('AMS_LOG_DIR', os.path.join(tmp, 'log'))
    from ams_hip import testing
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Trainer
    import utils.ops
    utils.ops.rng.seed(42)          # the reference's module-level Conv1D RNG (utils/ops.py:5): make the init order-independent here
    folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre'), window_size=W, filters=N, hop_size=hop,
                                                   chunk_size=L, batch_size=B, nb_speakers=S)
    args = dict(params)
    args.update(testing.SEPARATOR_DEFAULTS)
    args.update(layer_size=layer_size, nb_layers=nb_layers, embedding_size=E, model_folder=folder, model_previous=None,
                batch_size=B, learning_rate=lr, optimizer=optimizer, pretraining=False)
    args.update(extra)
    trainer = Front_Separator_Trainer(DPCL, 'front_DPCL', **args)
    dist, tfds = trainer.prepare()
    return trainer, tfds


def run_smoke(torch, np, verbose=True):
    from oracle import step as ostep, optim as ooptim
    tmp = tempfile.mkdtemp(prefix='ams_smoke_')
    B, L, W, N, hop, LS, NL, E, S = 3, 1024, 64, 16, 16, 16, 2, 8, 2
    trainer, tfds = build_front_dpcl(tmp, B, L, W, N, hop, LS, NL, E, S)
    g = trainer.graph
    model = trainer.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        P = {n: v.detach().cpu().numpy().astype(np.float64) for n, v in g.variables.items()}
        cost = float(model.train(feed, 0))
        run = model.last_run
        x_mix = model.x_mix.value(run).cpu().numpy().astype(np.float64)
        x_nm = model.x_non_mix.value(run).cpu().numpy().astype(np.float64)
        V = model.sepNet.prediction.value(run).detach().cpu().numpy()
        grads = {v.ams_name: v.grad.detach().cpu().numpy() for v in model.trainable_variables}
        P_new = {v.ams_name: v.detach().cpu().numpy() for v in model.trainable_variables}
    c_ref, g_ref, V_ref, Y_ref = ostep.front_dpcl_loss(x_mix, x_nm, P, hop, NL, E)

    def rel(a, b):
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    errs = {'cost': abs(cost - c_ref) / max(abs(c_ref), 1e-30), 'embeddings': rel(V, V_ref)}
    names = sorted(g_ref)
    for n in names:
        errs['grad ' + n] = rel(grads[n], g_ref[n])
    opt = ooptim.AMSGrad(1e-3)
    plist = [P[n].copy() for n in names]
    opt.apply(plist, [g_ref[n] for n in names])
    for n, p in zip(names, plist):
        errs['update ' + n] = rel(P_new[n], p)
    worst = max(errs.values())
    if verbose:
        print('smoke: front_DPCL step cost=%.6f (oracle %.6f); worst rel err %.2e over %d checks' % (cost, c_ref, worst, len(errs)))
    assert worst < 1e-3, errs                      # north_star tolerance: 1e-3 relative fp32
    return errs
