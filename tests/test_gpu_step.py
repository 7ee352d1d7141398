"""GPU: whole training steps through the host mirror (reference API) vs the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def test_front_dpcl_training_step_matches_oracle():
    from tests.smoke_step import run_smoke
    errs = run_smoke(torch, np, verbose=False)
    assert max(errs.values()) < 2e-4, errs


def test_hip_graph_replay_matches_eager():
    """--hip_graph: 2 eager warm-up steps, capture, then replays -- parameters after 5 steps equal the all-eager run
    (same kernels, same accumulation order; the weight-gradient side stream is part of the captured graph)."""
    import tempfile
    from tests.smoke_step import build_front_dpcl

    def run(graph):
        tmp = tempfile.mkdtemp(prefix='ams_cg_')
        trainer, tfds = build_front_dpcl(tmp, B=4, L=1024, W=64, N=16, hop=16, layer_size=16, nb_layers=2, E=8, hip_graph=graph,
                                         no_summaries=True)
        g, model = trainer.graph, trainer.model
        costs = []
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: 1024}
            tfds.initialize(tfds.TRAIN)
            for i in range(5):
                costs.append(float(model.train(feed, i)))
        torch.cuda.synchronize()
        return costs, {v.ams_name: v.detach().cpu().numpy().copy() for v in model.trainable_variables}

    c_e, p_e = run(False)
    c_g, p_g = run(True)
    assert np.allclose(c_e, c_g, rtol=1e-6, atol=0), (c_e, c_g)
    for n in p_e:
        assert np.array_equal(p_e[n], p_g[n]) or np.abs(p_e[n] - p_g[n]).max() <= 1e-7 * max(1.0, np.abs(p_e[n]).max()), n
