"""CPU: host logic of the reference-API mirror -- CLI surface, graph construction and variable naming,
freeze/restore semantics, params JSON, checkpoint layout.  No kernel is launched."""
import json
import os
import sys

import numpy as np
import pytest

os.environ.setdefault('AMS_LOG_DIR', '/tmp/ams_test_log')
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'adaptive-multispeaker-separation_amd')


def _parse(script_groups, argv):
    from utils.trainer import MyArgs
    p = MyArgs()
    for g in script_groups:
        if g == 'model_folder':
            p.parser.add_argument('--model_folder', required=True)
        elif g == 'model_folder_opt':
            p.parser.add_argument('--model_folder', required=False, default=None)
        else:
            getattr(p, g)()
    return p.get_args(argv)


def test_cli_parses_reference_command_lines():
    # README.md:23
    a = _parse(['add_adapt_args'], '--men --women --loss sdr+l2 --separation mask --learning_rate 0.001 --nb_speakers 2 '
               '--batch_size 4 --filters 256 --max_pool 256 --beta 0.0 --regularization 0.0 --overlap_coef 1.0 '
               '--no_random_picking'.split())
    assert a.filters == 256 and a.loss == 'sdr+l2' and a.sex == ['M', 'F'] and a.with_max_pool is False
    assert a.window_size == 1024 and a.hop_size == 256 and a.chunk_size == 20480
    # dpcl_stft_train.sh:24-26 flags
    a = _parse(['model_folder_opt', 'add_stft_args', 'add_separator_args'],
               '--dataset h5py_files/train-clean-100-8-s.h5 --chunk_size 20480 --nb_speakers 2 --epochs 10 --batch_size 124 '
               '--learning_rate 0.001 --window_size 512 --hop_size 256 --layer_size 600 --embedding_size 40 --men --women'.split())
    assert a.window_size == 512 and a.model_folder is None and a.no_normalize is True and a.nb_tries == 10
    assert a.beta_kmeans is None and a.optimizer == 'Adam' and a.decay_epoch == 50
    # front_dpcl_finetuning.pbs:25-32
    a = _parse(['model_folder', 'add_adapt_args', 'add_separator_args', 'add_finetuning_args', 'add_enhance_layer_args'],
               '--model_folder log/front_DPCL_enhance/x --nb_speakers 2 --epochs 100 --batch_size 32 --chunk_size 10240 '
               '--nb_tries 1 --nb_steps 10 --beta_kmeans 10.0 --with_silence --threshold 2.0 --end_assign --learning_rate 0.001 '
               '--optimizer RMSProp --train prediction enhance --men --no_random_picking'.split())
    assert a.beta_kmeans == 10.0 and a.train == ['prediction', 'enhance'] and a.end_assign and a.nonlinearity == 'softmax'


def test_every_entry_point_exists_and_imports():
    names = ['pretraining', 'STFT_DPCL', 'STFT_L41', 'STFT_DPCL_enhance', 'STFT_L41_enhance', 'STFT_DPCL_finetuning',
             'STFT_L41_finetuning', 'front_DPCL', 'front_L41', 'front_DPCL_enhance', 'front_L41_enhance',
             'front_DPCL_finetuning', 'front_L41_finetuning', 'front_DPCL_enhance_finetuning', 'front_L41_enhance_finetuning']
    import importlib
    for n in names:
        assert os.path.exists(os.path.join(PKG, 'experiments', 'training', n + '.py')), n
        importlib.import_module('experiments.training.' + n)


def test_entry_point_table_registers_the_reference_flags():
    """experiments/training/_recipes.py: per script the Trainer class, the separator, the `type` string and the flag groups of the
    reference's entry points (experiments/training/*.py)."""
    from experiments.training import _recipes as R
    import utils.trainer as T
    assert set(R.RECIPES) == {'pretraining', 'STFT_DPCL', 'STFT_L41', 'STFT_DPCL_enhance', 'STFT_L41_enhance', 'STFT_DPCL_finetuning',
                              'STFT_L41_finetuning', 'front_DPCL', 'front_L41', 'front_DPCL_enhance', 'front_L41_enhance',
                              'front_DPCL_finetuning', 'front_L41_finetuning', 'front_DPCL_enhance_finetuning',
                              'front_L41_enhance_finetuning'}
    for name, (trainer, sep, typ, need_folder, has_prev, groups, pre) in R.RECIPES.items():
        assert hasattr(T, trainer), trainer
        argv = ['--men', '--women', '--nb_speakers', '2']
        if need_folder:
            argv += ['--model_folder', 'log/x']
        a = R.build_parser(name).get_args(argv)
        assert a.sex == ['M', 'F'] and a.nb_speakers == 2
        assert hasattr(a, 'model_folder') == (need_folder is not None)
        assert hasattr(a, 'model_previous') == has_prev
        assert hasattr(a, 'nb_tries') == ('separator' in groups) and hasattr(a, 'train') == ('finetuning' in groups)
        assert hasattr(a, 'nonlinearity') == ('enhance_layer' in groups)
        assert hasattr(a, 'filters') == ('adapt' in groups)
        if 'stft' in groups:
            assert a.window_size == 512 and a.hop_size == 256
        elif 'adapt' in groups:
            assert a.window_size == 1024
    # the reference quirks kept: STFT_{DPCL,L41} do not require --model_folder; the DPCL fine-tuning script passes 'front_L41_finetuning'
    assert R.RECIPES['STFT_DPCL'][3] is False and R.RECIPES['front_DPCL_finetuning'][2] == 'front_L41_finetuning'
    with pytest.raises(SystemExit):
        R.build_parser('front_DPCL').get_args(['--men'])             # --model_folder is required there


# ---- the CLI boundary against the reference itself: tests/golden/cli.json is what tests/golden/make_cli_golden.py read off
# /root/reference/utils/trainer.py:10-176 (MyArgs executed in place) and experiments/training/*.py (ast) ----

AMS_FLAGS = {'--synthetic_batches', '--synthetic_pool', '--no_summaries', '--hip_graph', '--f16_audit_every', '--kmeans_seeding'}
# reference scripts that hang off models outside SURVEY 8 (models/SC_V2.py, models/focus.py, models/enhanced_L41.py): not built
OUT_OF_SCOPE_SCRIPTS = {'STFT_L41V2': 'models.SC_V2', 'front_L41V2': 'models.SC_V2', 'front_focus': 'models.focus',
                        'front_mm': 'models.enhanced_L41'}


def _cli_golden():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cli.json')) as f:
        return json.load(f)


def _describe(action):
    return {'flags': list(action.option_strings), 'dest': action.dest, 'kind': type(action).__name__.lstrip('_'),
            'type': action.type.__name__ if action.type is not None else None, 'default': action.default,
            'required': bool(action.required), 'choices': list(action.choices) if action.choices is not None else None,
            'nargs': action.nargs}


def test_myargs_registers_exactly_the_reference_flags():
    """Every flag of the reference's MyArgs -- name, dest, type, default, required, choices, nargs, store / store_true / store_false --
    in the base parser and in each add_*_args group, in the reference's order; the only extras are this implementation's six
    documented additions in the base parser."""
    import argparse
    from utils.trainer import MyArgs
    gold = _cli_golden()['myargs']
    base = [_describe(a) for a in MyArgs().parser._actions if not isinstance(a, argparse._HelpAction)]
    extras = [a for a in base if a['flags'][0] in AMS_FLAGS]
    assert [a for a in base if a['flags'][0] not in AMS_FLAGS] == gold['base']
    assert {a['flags'][0] for a in extras} == AMS_FLAGS and not any(a['required'] for a in extras)
    for method, want in gold['groups'].items():
        p = MyArgs()
        n0 = len(p.parser._actions)
        getattr(p, method)()
        assert [_describe(a) for a in p.parser._actions[n0:]] == want, method
    for argv, sex in gold['sex'].items():
        assert MyArgs().get_args(argv.split()).sex == sex


def test_entry_points_are_the_reference_scripts():
    """Per entry script of the reference: Trainer class, separator class, `type` string, the pretraining keyword, the argument groups
    in order, --model_folder / --model_previous with their `required`.  Scripts of the reference that are absent here are exactly the
    four that import a model outside the hot path."""
    from experiments.training import _recipes as R
    gold = _cli_golden()['scripts']
    assert set(gold) - set(R.RECIPES) == set(OUT_OF_SCOPE_SCRIPTS) and not set(R.RECIPES) - set(gold)
    for name, module in OUT_OF_SCOPE_SCRIPTS.items():
        assert gold[name]['separator_import']['module'] == module
    for name, (trainer, sep, typ, need_folder, has_prev, groups, pre) in R.RECIPES.items():
        g = gold[name]
        assert g['construct']['trainer'] == trainer and g['calls'] == ['train'], name
        assert g['construct']['args'] == ([] if sep is None else [sep, typ]), name
        assert g['construct']['kwargs'] == ({} if pre is None else {'pretraining': pre}), name
        assert (g['separator_import'] or {'names': [None]})['names'] == [sep], name
        assert tuple(m[len('add_'):-len('_args')] for m in g['groups']) == groups, name
        inline = {f['flags'][0]: f for f in g['inline_flags']}
        assert set(inline) == ({'--model_folder'} if need_folder is not None else set()) | ({'--model_previous'} if has_prev else set()), name
        if need_folder is not None:
            assert inline['--model_folder'].get('required', False) == need_folder, name
        # and the parser this table builds says the same
        acts = {a.option_strings[0]: a for a in R.build_parser(name).parser._actions if a.option_strings}
        for flag, f in inline.items():
            assert acts[flag].required == f.get('required', False) and acts[flag].default == f.get('default'), (name, flag)


def test_trainer_classes_are_the_reference_classes():
    import utils.trainer as T
    gold = _cli_golden()['trainer_classes']
    not_on_path = {'Adapt_Enhance', 'MultiChannel_Pretrainer'}        # utils/trainer.py:539-569: an undefined MultiAdapt / SURVEY 2 row 15
    for name, c in gold.items():
        if name in not_on_path:
            assert not hasattr(T, name)
            continue
        cls = getattr(T, name)
        assert cls.__mro__[1].__name__ == c['base'], name
        for m in c['methods']:
            assert callable(getattr(cls, m)), (name, m)


def test_front_dpcl_construction_names_and_freeze(tmp_path):
    from tests.smoke_step import build_front_dpcl
    trainer, tfds = build_front_dpcl(str(tmp_path), B=2, L=256, W=32, N=8, hop=8, layer_size=8, nb_layers=2, E=4)
    g, model = trainer.graph, trainer.model
    names = list(g.variables)
    assert names[:2] == ['front/window/w', 'front/bases/bases']
    for n in ['prediction/forward_BLSTM_0/rnn/basic_lstm_cell/kernel', 'prediction/backward_BLSTM_1/rnn/basic_lstm_cell/bias',
              'prediction/W', 'prediction/b', 'back/window/value', 'back/bases/value']:
        assert n in g.variables, n
    assert g.variables['prediction/forward_BLSTM_0/rnn/basic_lstm_cell/kernel'].shape == (8 + 4, 16)
    assert g.variables['prediction/forward_BLSTM_1/rnn/basic_lstm_cell/kernel'].shape == (8 + 4, 16)
    assert g.variables['prediction/W'].shape == (8, 4 * 8)
    trainable = [v.ams_name for v in model.trainable_variables]
    assert all(n.startswith('prediction/') for n in trainable) and len(trainable) == 2 * 2 * 2 + 2
    # front/back restored from the checkpoint, the rest reported as non-initialised then initialised
    assert {'front/window/w', 'back/bases/value'} <= g.initialized and not model.non_initialized_variables()
    # trainable variables are views into one flat buffer (the all-reduce / fused optimizer buffer)
    opt = model.optimize
    assert opt.flat.numel() == sum(v.numel() for v in model.trainable_variables)
    v0 = model.trainable_variables[0]
    assert v0.data_ptr() == opt.flat.data_ptr() and v0.grad.data_ptr() == opt.flat_grad.data_ptr()
    assert not g.variables['front/window/w'].requires_grad and v0.requires_grad
    # params JSON written beside the run, checkpoint round trip
    run_dir = model._dir()
    params = json.load(open(os.path.join(run_dir, 'params')))
    assert params['type'] == 'front_DPCL' and params['filters'] == 8 and 'mix' not in params
    with g.as_default():
        model.save(7)
        before = g.variables['prediction/W'].detach().clone()
        g.variables['prediction/W'].data.add_(1.0)
        model.restore_last_checkpoint()
        assert np.array_equal(before.numpy(), g.variables['prediction/W'].detach().numpy())
    assert os.path.exists(os.path.join(run_dir, 'model-7.npz'))


def test_synthetic_source_contract():
    from data.dataset import synthetic_mixtures
    mix, nm, ind = synthetic_mixtures([0, 1, 5], 2, 2048)
    assert mix.shape == (3, 2048) and nm.shape == (3, 2, 2048) and ind.shape == (3, 2)
    assert np.allclose(mix, nm.sum(1)) and (ind[:, 0] != ind[:, 1]).all() and ind.max() < 251
    assert abs(np.sqrt((nm[0, 0] ** 2).mean()) - 0.05) < 1e-6
    m2, _, _ = synthetic_mixtures([1], 2, 2048)
    assert np.array_equal(m2[0], mix[1])                   # depends on the global utterance index only


def test_kmeans_reference_seeding():
    """Default k-means restarts ARE models/Kmeans_2.py:61-66: one np.random.choice(range(l), size=C, replace=False) per row, in row
    order, from the global numpy RNG seeded 42 (models/network.py:17-18) -- same indices, same stream position afterwards."""
    import types
    import numpy as np
    from ams_hip.kmeans_host import KMeans
    R, L, C = 12, 500, 3
    np.random.seed(42)
    ref = np.array([np.random.choice(range(L), size=C, replace=False) for _ in range(R)]).astype(np.int32)
    after_ref = np.random.randint(0, 1 << 30)
    np.random.seed(42)
    got = KMeans._draw(types.SimpleNamespace(nb_clusters=C, seeding='reference'), R, L).numpy()
    after_got = np.random.randint(0, 1 << 30)
    assert got.dtype == np.int32 and np.array_equal(got, ref) and after_got == after_ref
    fast = KMeans._draw(types.SimpleNamespace(nb_clusters=C, seeding='fast'), 200, 50).numpy()
    assert fast.shape == (200, C) and all(len(set(r)) == C for r in fast.tolist()) and fast.min() >= 0 and fast.max() < 50


def test_kmeans_reference_seeding_shapes_and_stream_sharing():
    """The C draw (csrc/host/mt_choice.c) against numpy itself over the shapes the recipes use and the edge ones (l = 1, C = l,
    powers of two either side, l > 65536), from odd stream positions; and the one-batch-ahead speculation: used when nobody
    else touched the global generator, dropped -- without changing what anyone sees -- when somebody did."""
    import types
    import numpy as np
    from ams_hip import kmeans_host as kh
    assert kh._REFERENCE_SEEDS._native(), 'libams_host.so must be built (make -C adaptive-multispeaker-separation_amd/csrc)'
    for (R, L, C) in [(5, 1, 1), (7, 2, 2), (9, 3, 3), (20, 4, 3), (20, 5, 2), (20, 8, 8), (20, 9, 1), (50, 17, 17), (6, 1024, 2),
                      (6, 1025, 5), (4, 20480, 2), (2, 40960, 3), (2, 20303, 2), (3, 65537, 4)]:
        ns = types.SimpleNamespace(nb_clusters=C, seeding='reference')
        for seed in (42, 7):
            np.random.seed(seed)
            np.random.randint(0, 10, size=seed)
            ref = np.array([np.random.choice(range(L), size=C, replace=False) for _ in range(R)]).astype(np.int32)
            after_ref = np.random.randint(0, 1 << 30)
            np.random.seed(seed)
            np.random.randint(0, 10, size=seed)
            got = kh.KMeans._draw(ns, R, L).numpy()
            assert np.array_equal(ref, got) and np.random.randint(0, 1 << 30) == after_ref, (R, L, C, seed)
    ns = types.SimpleNamespace(nb_clusters=3, seeding='reference')
    np.random.seed(1)
    r = [np.array([np.random.choice(range(300), size=3, replace=False) for _ in range(5)]) for _ in range(3)]
    x = np.random.normal()                                  # leaves a cached gaussian in the state as well
    r.append(np.array([np.random.choice(range(300), size=3, replace=False) for _ in range(5)]))
    x2 = np.random.normal()
    np.random.seed(1)
    hits = kh._REFERENCE_SEEDS.hits
    g = [kh.KMeans._draw(ns, 5, 300).numpy() for _ in range(3)]
    assert kh._REFERENCE_SEEDS.hits - hits == 2             # batches 2 and 3 came from the worker thread
    y = np.random.normal()
    g.append(kh.KMeans._draw(ns, 5, 300).numpy())           # the speculation started before the foreign draw: dropped
    assert kh._REFERENCE_SEEDS.hits - hits == 2
    assert all(np.array_equal(a, b) for a, b in zip(r, g)) and x == y and np.random.normal() == x2
    with pytest.raises(ValueError):                         # numpy's own error for C > l
        kh.KMeans._draw(types.SimpleNamespace(nb_clusters=4, seeding='reference'), 2, 3)


def test_tf_eval_running_mean_and_model_choices():
    """experiments/evaluation/tf_eval.py:27-38: batch-size-weighted running mean of the per-batch in-graph SDR improvement, NaN
    batches skipped; --model choices the reference leaves without an inferencer exit instead of crashing with a NameError."""
    import numpy as np
    import pytest
    from experiments.evaluation import tf_eval
    batches = [(None, None, np.float32(3.0)), (None, None, np.float32('nan')), (None, None, np.array([5.0], np.float32))]
    sdr, n = tf_eval.running_sdr(batches, batch_size=4, verbose=False)
    assert n == 2 and abs(sdr - 4.0) < 1e-6
    assert np.isnan(tf_eval.running_sdr([], 4, verbose=False)[0])
    assert set(tf_eval.INFERENCERS) == {'front_L41', 'STFT_L41', 'front_L41_enhance', 'pretraining'}
    with pytest.raises(SystemExit):
        tf_eval.main(['--model_folder', '/nonexistent', '--model', 'front_L41_finetuned'])
    with pytest.raises(SystemExit):                                       # --model is required (utils/trainer.py:112-117)
        tf_eval.main(['--model_folder', '/nonexistent'])


def test_keyed_kmeans_seeds_do_not_depend_on_the_number_of_ranks():
    """SURVEY 8e "Partitioning": with G ranks the k-means restarts of utterance j must not depend on G.  'keyed' seeding (what
    ams_hip.kmeans_host.KMeans uses under data parallelism instead of the row-by-row host stream of Kmeans_2.py:61-66) indexes a
    counter-based stream by (draw, GLOBAL row): the rows of two half-size shards are the rows of the whole batch; C distinct bins."""
    from ams_hip.kmeans_host import KMeans, keyed_seeds
    from ams_hip.graph import Graph
    whole = keyed_seeds(5, 0, 64 * 10, 20480, 2)
    halves = np.concatenate([keyed_seeds(5, 0, 32 * 10, 20480, 2), keyed_seeds(5, 32 * 10, 32 * 10, 20480, 2)])
    assert np.array_equal(whole, halves)
    assert whole.min() >= 0 and whole.max() < 20480 and (whole[:, 0] != whole[:, 1]).all()
    assert not np.array_equal(whole, keyed_seeds(6, 0, 64 * 10, 20480, 2))          # the next draw is another one
    three = keyed_seeds(0, 0, 5000, 7, 3)                                           # small L: collisions are redrawn
    assert all(len(set(r)) == 3 for r in three.tolist())

    class _D(object):
        enabled, world_size = True, 2
        rows = (320, 320)                     # what the two ranks hold in the current draw

        def __init__(self, rank):
            self.rank = rank
            self.draws = 0

        def all_gather_object(self, obj):     # stands in for the process group: every rank reports (rows, draw number)
            assert obj == (self.rows[self.rank], self.draws)
            self.draws += 1
            return [(r, obj[1]) for r in self.rows]
    with Graph().as_default():
        kms = [KMeans(2, nb_tries=10, dist=_D(r)) for r in (0, 1)]
        one = KMeans(2, nb_tries=10, seeding='keyed')
    assert kms[0].seeding == 'keyed'                                                # 'reference' is a single-process stream
    for step in range(2):
        parts = [km._draw(320, 20480).numpy() for km in kms]
        assert np.array_equal(np.concatenate(parts), one._draw(640, 20480).numpy())
    # a ragged last batch: the ranks hold 320 and 170 rows -- the second rank's rows follow the first's, none shared, none skipped
    _D.rows = (320, 170)
    parts = [km._draw(r, 20480).numpy() for km, r in zip(kms, _D.rows)]
    assert np.array_equal(np.concatenate(parts), one._draw(490, 20480).numpy())


def test_the_forward_product_form_is_decided_by_the_settled_part_of_each_tuning_block(monkeypatch):
    """models/network.py::_ps_finish_tuning: a captured step replays its two forms in alternating blocks; only the LAST THIRD of a block
    counts (the clock governor needs tens of milliseconds after the form changes: blocks of 8 steps measured both forms at the slower
    form's clock and chose wrongly), the medians decide, the other graph is dropped and the decision stays on the model."""
    import torch
    from models.network import Network
    from ams_hip import ops

    class Ev(object):
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    monkeypatch.setattr(Network, '_PS_TUNE_BLOCK', 12)
    net = Network.__new__(Network)

    def events(settled, unsettled):
        # steps 0..7 of a block run at the previous form's clock (`unsettled`), steps 8..11 at this form's own
        return [(i, Ev(0.0), Ev(unsettled if i < 8 else settled)) for i in range(12)] * 2
    st = {'tune': {'variants': [('gA', 'cA', 'rA'), ('gB', 'cB', 'rB')], 'k': 48, 'ev': (events(2.63, 2.80), events(2.76, 2.60))}}
    ops.PS_TUNED.clear()
    net._ps_finish_tuning(st)
    assert st['graph'] == 'gA' and st['tune'] is None and net._ps_choice is True          # pre-split wins on its settled steps
    assert ops.PS_TUNED['presplit'] is True and abs(ops.PS_TUNED['ms_presplit'] - 2.63) < 1e-9 and ops.PS_TUNED['steps_counted'] == 8
    st = {'tune': {'variants': [('gA', 'cA', 'rA'), ('gB', 'cB', 'rB')], 'k': 48, 'ev': (events(2.81, 2.70), events(2.77, 2.90))}}
    net._ps_finish_tuning(st)
    assert st['graph'] == 'gB' and net._ps_choice is False and ops.PS_TUNED['presplit'] is False
