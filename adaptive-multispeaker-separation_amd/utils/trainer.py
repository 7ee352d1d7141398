# coding: utf-8
"""Argument layer + training recipes (reference utils/trainer.py), host mirror.

``MyArgs`` keeps the reference's flag names and defaults; ``Trainer.train`` keeps its
loop (train -> every validation_step validate -> save-if-best -> epoch-end lr decay -> final validation,
restore best, test); the 13 recipe classes (same names and constructor signatures; the reference's two `Adapt_Enhance` / `MultiChannel_Pretrainer` classes are not on the path) are generated from the ``WIRING`` table
below, which states what each one builds, restores, freezes and optimises.  Additions (all optional): ``--synthetic_batches``,
``--synthetic_pool`` (synthetic data source size), ``--no_summaries``, ``--hip_graph``; data parallelism is picked up from
torchrun's environment (WORLD_SIZE/RANK/LOCAL_RANK), one process per GPU, RCCL all-reduce of gradients.
"""
from __future__ import print_function

import argparse
import os
import time

import numpy as np
import torch

from ams_hip import ops
from ams_hip.dist import Dist
from ams_hip.graph import Graph
from data.dataset import TFDataset
from models.adapt import Adapt
from utils.tools import getETA


class MyArgs(object):

    def __init__(self):
        parser = argparse.ArgumentParser(description="Argument Parser")

        # DataSet arguments
        parser.add_argument('--dataset', help='Path to H5 dataset from workspace', required=False,
                            default='h5py_files/train-clean-100-8-s.h5')
        parser.add_argument('--dataset_normalize', help='Mean/Std normalization for each input', action="store_true")
        parser.add_argument('--chunk_size', type=int, help='Chunk size for inputs', required=False, default=20480)
        parser.add_argument('--nb_speakers', type=int, help='Number of mixed speakers', required=False, default=2)
        parser.add_argument('--no_random_picking', help='Do not pick random genders when mixing', action="store_true")
        parser.add_argument('--validation_step', type=int, help='Nb of steps between each validation', required=False,
                            default=1000)
        parser.add_argument('--men', help='Use men voices', action="store_true")
        parser.add_argument('--women', help='Use women voices', action="store_true")

        # Training arguments
        parser.add_argument('--epochs', type=int, help='Number of epochs', required=False, default=10)
        parser.add_argument('--batch_size', type=int, help='Batch size', required=False, default=64)
        parser.add_argument('--learning_rate', type=float, help='learning rate for training', required=False, default=0.1)
        parser.add_argument('--optimizer', help='Optimizer used during training', choices=['Adam', 'SGD', 'RMSProp'],
                            required=False, default='Adam')
        parser.add_argument('--decay_epoch', type=int, help='Number of epoch to apply learning rate decay', required=False,
                            default=50)
        parser.add_argument('--gradient_norm_clip', type=float, help='Clip the gradient norm by this value if != 0',
                            required=False, default=0.)

        # Additions of this implementation (synthetic data source; logging)
        parser.add_argument('--synthetic_batches', type=int, help='[ams] batches per epoch of the synthetic source',
                            required=False, default=20)
        parser.add_argument('--synthetic_pool', type=int, help='[ams] distinct synthetic batches kept on device',
                            required=False, default=8)
        parser.add_argument('--no_summaries', help='[ams] do not write per-step summaries', action="store_true")
        parser.add_argument('--hip_graph', help='[ams] capture forward+backward of the training step into a hipGraph after two '
                            'eager steps and replay it (fixed batch shape; the 480 recurrent launches of a 3xBLSTM step become '
                            'one graph launch)', action="store_true")
        parser.add_argument('--f16_audit_every', type=int, default=1000, help='[ams] every N steps (and at step 10) one training step runs '
                            'eagerly with the fp16x3 operand-range audit on: a product class whose operands leave the fp16 range falls '
                            'back to bf16x6 (0 = never)')
        parser.add_argument('--kmeans_seeding', choices=['reference', 'fast', 'keyed'], default='reference',
                            help="[ams] k-means restarts: 'reference' = one np.random.choice per row exactly as models/Kmeans_2.py:61-66 "
                            "(bit-exact index stream, serial host draw); 'fast' = one vectorised draw (same distribution, other stream); 'keyed' = a "
                            "counter-based stream indexed by the GLOBAL row (what N > 1 ranks use instead of 'reference': independent of N)")
        self.parser = parser

    def add_stft_args(self):
        self.parser.add_argument('--window_size', type=int, help='Size of the window for STFT', required=False, default=512)
        self.parser.add_argument('--hop_size', type=int, help='Hop size for the STFT', required=False, default=256)

    def add_separator_args(self):
        p = self.parser
        # Preprocessing parameters
        p.add_argument('--normalize_separator', help='Normalize the input of the separator',
                       choices=['None', '01', 'meanstd'], required=False, default='None')
        p.add_argument('--abs_input', help='Abs on the separator input', action="store_true")
        p.add_argument('--pre_func', help='#TODO', choices=['None', 'sqrt', 'log'], required=False, default='None')
        p.add_argument('--silence_mask_db', type=int, help='silence mask applied to the input under this threshold',
                       required=False, default=0)
        # Architecture params
        p.add_argument('--nb_layers', type=int, help='Number of stacked BLSTMs', required=False, default=3)
        p.add_argument('--layer_size', type=int, help='Size of hidden layers in BLSTM', required=False, default=600)
        p.add_argument('--embedding_size', type=int, help='Size of the embedding output', required=False, default=40)
        p.add_argument('--no_normalize', help='Normalization of the embedded space', action="store_false")
        p.add_argument('--recurrent_dropout', type=float, help='Dropout for the recurrent layers', required=False, default=0.0)
        # KMEANS PARAMS
        p.add_argument('--nb_tries', type=int, help='Number of tries for KMEANS', required=False, default=10)
        p.add_argument('--nb_steps', type=int, help='Number of steps for KMEANS', required=False, default=10)
        p.add_argument('--beta_kmeans', type=float, help='Beta value for KMEANS - None = Hard KMEANS', required=False,
                       default=None)
        p.add_argument('--threshold', type=float, help='Threshold for the silent bins', required=False, default=2.0)
        p.add_argument('--with_silence', help='Silence weak bins during KMEANS', action="store_true")
        p.add_argument('--end_assign', help='Assign the silent bins', action="store_true")
        # L41 Loss params
        p.add_argument('--silence_loss', help='Silence weak bins in the loss function', action="store_true")
        p.add_argument('--threshold_silence_loss', type=float, help='Threshold for the silent bins', required=False,
                       default=2.0)
        p.add_argument('--function_mask', help='#TODO', choices=['None', 'linear', 'sqrt', 'square'], required=False,
                       default='None')
        p.add_argument('--sampling', type=int, help='#TODO', required=False, default=None)
        p.add_argument('--ns_rate', type=float, help='#TODO', required=False, default=0.1)
        p.add_argument('--ns_method', help='#TODO', choices=['random', 'k-nearest'], required=False, default='random')
        # Adding new architectures
        p.add_argument('--add_dilated', help='Add Convolutional Dilated Network before the BLSTMs', action="store_true")

    def select_inferencer(self):
        self.parser.add_argument('--model', help='#TODO',
                                 choices=['pretraining', 'front_L41', 'front_L41_finetuned', 'front_L41_enhance',
                                          'front_L41_enhanced_finetuned', 'STFT_L41', 'STFT_L41_finetuned',
                                          'STFT_L41_enhanced', 'STFT_L41_enhanced_finetuned'], required=True)

    def add_finetuning_args(self):
        self.parser.add_argument('--train', nargs='*')

    def add_enhance_layer_args(self):
        p = self.parser
        p.add_argument('--normalize_enhance', help='Normalize the input of the enhance layer', action="store_true")
        p.add_argument('--nb_layers_enhance', type=int, help='Number of stacked BLSTMs for the enhance layer',
                       required=False, default=3)
        p.add_argument('--layer_size_enhance', type=int, help='Size of hidden layers in BLSTM', required=False, default=600)
        p.add_argument('--nonlinearity', help='Nonlinearity used in output', choices=['tanh', 'softmax', 'None'],
                       required=False, default='softmax')
        p.add_argument('--recurrent_dropout_enhance', type=float, help='Dropout for the recurrent layers', required=False,
                       default=0.0)

    def add_adapt_args(self):
        p = self.parser
        # Preprocess arguments
        p.add_argument('--window_size', type=int, help='Size of the 1D Conv width', required=False, default=1024)
        p.add_argument('--filters', type=int, help='Number of filters/bases for the 1D Conv', required=False, default=512)
        p.add_argument('--max_pool', type=int, help='Max Pooling size', required=False, default=512)
        p.add_argument('--with_max_pool', help='Use Max pooling and not hop', action="store_true")
        p.add_argument('--with_average_pool', help='Use Average pooling and not hop', action="store_true")
        p.add_argument('--hop_size', type=int, help='Hop size for the STFT', required=False, default=256)
        # Loss arguments
        p.add_argument('--regularization', type=float, help='Coefficient for L2 regularization', required=False, default=1e-4)
        p.add_argument('--beta', type=float, help='Coefficient for Sparsity constraint', required=False, default=1e-2)
        p.add_argument('--sparsity', type=float, help='Average Sparsity constraint', required=False, default=0.01)
        p.add_argument('--overlap_coef', type=float, help='Coefficient for Overlapping loss', required=False, default=0.001)
        p.add_argument('--overlap_value', type=float, help='Coefficient for Overlapping loss', required=False, default=0.1)
        p.add_argument('--non_negativity', type=float, help='Coefficient for Non-Negativity loss', required=False, default=0.0)
        p.add_argument('--loss', choices=['l2', 'sdr', 'l2+sdr', 'sdr+l2'], required=False, default='sdr')
        p.add_argument('--separation', choices=['perfect', 'mask'], required=False, default='perfect')

    def get_args(self, argv=None):
        parsed = self.parser.parse_args(argv)
        sex = []
        if parsed.men:
            sex.append('M')
        if parsed.women:
            sex.append('F')
        parsed.sex = sex
        return parsed


class Trainer(object):
    def __init__(self, trainer_type, **kwargs):
        self.batch_size = kwargs['batch_size']
        additional_args = {"type": trainer_type}
        kwargs.update(additional_args)
        self.args = kwargs

    # ------------------------------------------------------------------ plumbing shared by train/inference
    def _open(self):
        dist = self.args.get('dist') or Dist()
        self.args['dist'] = dist
        self.args['summaries'] = not self.args.get('no_summaries', False)
        device = torch.device('cuda', dist.local_rank) if torch.cuda.is_available() else torch.device('cpu')
        if device.type == 'cuda':
            torch.cuda.set_device(device)
        self.graph = Graph(device)
        return dist

    def _pipeline_args(self, tfds):
        return {"mix": tfds.next_mix, "non_mix": tfds.next_non_mix, "ind": tfds.next_ind, "pipeline": True,
                "tot_speakers": 251}

    def _sync_replicas(self, dist):
        """Identical weights on every rank: broadcast rank 0's variables."""
        if dist.enabled:
            for v in self.graph.global_variables():
                dist.broadcast(v.data)
            # the variables were written behind the model's back: whatever a pass derived from them and kept (bounds, images, gathered
            # kernels of variables nobody trains; a captured step's constants) is stale
            from models.network import Network
            Network._weights_written()

    def prepare(self):
        """Everything Trainer.train does before its loop: graph, input pipeline, build(), replica sync."""
        dist = self._open()
        with self.graph.as_default():
            tfds = TFDataset(**self.args)
            self.args.update(self._pipeline_args(tfds))
            self.build()
            self._sync_replicas(dist)
        self.tfds = tfds
        return dist, tfds

    def inference(self):
        dist = self._open()
        with self.graph.as_default():
            tfds = TFDataset(**self.args)
            self.args.update(self._pipeline_args(tfds))
            self.build()
            split = tfds.TEST_OTHER if self.args.get("out") else tfds.TEST
            nb_batches_test = tfds.length(split)
            feed_dict_test = {tfds.handle: tfds.get_handle(split), tfds.chunk_size: self.args['chunk_size']}
            tfds.initialize(split)
            for b in range(nb_batches_test):
                output = self.model.infer(feed_dict_test, b)
                print('Batch #', b + 1, '/', nb_batches_test, end=' ')
                yield output

    def sdr_improvement(self):
        dist = self._open()
        with self.graph.as_default():
            tfds = TFDataset(**self.args)
            self.args.update(self._pipeline_args(tfds))
            self.build()
            nb_batches_test = tfds.length(tfds.TEST)
            feed_dict_test = {tfds.handle: tfds.get_handle(tfds.TEST), tfds.chunk_size: self.args['chunk_size']}
            tfds.initialize(tfds.TEST)
            for b in range(nb_batches_test):
                output = self.model.improvement(feed_dict_test, b)
                yield output
                print('Batch #', b + 1, '/', nb_batches_test)

    # ------------------------------------------------------------------ the training loop (utils/trainer.py:262-390)
    def _feeds_for(self, tfds):
        chunk = self.args['chunk_size']
        return {split: {tfds.handle: tfds.get_handle(split), tfds.chunk_size: chunk}
                for split in (tfds.TRAIN, tfds.VALID, tfds.TEST)}

    def _validate(self, tfds, feeds, step):
        """One pass over the validation split; keeps the best model (save-if-best, :325-340 and :362-370)."""
        tfds.initialize(tfds.VALID)
        costs = [self.model.valid_batch(feeds[tfds.VALID], step) for _ in range(tfds.length(tfds.VALID))]
        mean = np.mean(costs)
        self.model.add_valid_summary(mean, step)
        if mean < self._best_cost:
            self._best_cost = mean
            self._best_path = self.model.save(step)
            self._say('Save best model with :', mean)
        return costs, mean

    def _say(self, *a, **k):
        if self._verbose:
            print(*a, **k)

    def train(self):
        dist, tfds = self.prepare()
        self._verbose = dist.rank == 0
        self._best_cost, self._best_path = 1e100, ''
        epochs, every = self.args['epochs'], self.args['validation_step']
        window = [0.0] * 10                        # moving average of the last 10 step times (:348-353)
        with self.graph.as_default():
            n_train, n_test = tfds.length(tfds.TRAIN), tfds.length(tfds.TEST)
            self._say('BATCHES')
            self._say(n_train, n_test, tfds.length(tfds.VALID))
            feeds = self._feeds_for(tfds)
            step, costs, mark = 0, [], time.time()
            for epoch in range(epochs):
                tfds.initialize(tfds.TRAIN)
                for b in range(n_train):
                    audit = int(self.args.get('f16_audit_every') or 0)
                    if audit > 0 and ops.F16X3 and (step == 10 or (step + 1) % audit == 0):
                        # run-time guard of the fp16x3 products: this step eagerly with the operand-range audit on (ops.F16_AUDIT)
                        c, denied = self.model.train_audited(feeds[tfds.TRAIN], step)
                        for k in denied:
                            self._say('fp16x3 audit: operands of product class', k, 'leave the fp16 range: bf16x6 from now on')
                    else:
                        c = self.model.train(feeds[tfds.TRAIN], step)
                    c = float(c)                   # host sync, as sess.run returning the cost does
                    if ops.ring_error_pending():
                        # a ring recurrence of this step could not get its workgroups resident in time: the optimizer skipped
                        # the update (the sticky error word is its guard); the same batch again on the per-step kernels.  Checked
                        # BEFORE any validation pass: an evaluation batch would see the word, clear it, and this step would be lost
                        # with the optimizer's host counters already advanced.
                        c = float(self.model.retrain_last(step))
                        self._say('recurrence ring gave up a bounded wait: step repeated on the per-step kernels')
                    if (step + 1) % every == 0:
                        t = time.time()
                        costs, mean = self._validate(tfds, feeds, step)
                        self._say('Validation set tested in ', time.time() - t, ' seconds')
                        self._say('Validation set: ', mean)
                    window = window[1:] + [time.time() - mark]
                    avg = sum(window) / len(window)
                    self._say('Epoch #', epoch + 1, '/', epochs, ' Batch #', b + 1, '/', n_train, 'in', avg, 'sec loss=', c,
                              ' ETA = ', getETA(avg, n_train, b + 1, epochs, epoch + 1))
                    mark = time.time()
                    step += 1
                self.model.increment_epoch()       # lr halves every decay_epoch epochs (network.py:171-177)

            costs, _ = self._validate(tfds, feeds, step)          # validation at the last step
            self._say('Best model with Validation:  ', self._best_cost)
            self._say('Path = ', self._best_path)

            # best model on the validation set -> test split.  The reference appends the test costs to the LAST validation
            # list before averaging (quirk C-13, :378-387); kept.
            dist.barrier()
            self.model.restore_last_checkpoint()
            tfds.initialize(tfds.TEST)
            costs = list(costs) + [self.model.test_batch(feeds[tfds.TEST]) for _ in range(n_test)]
            self.last_test_cost = float(np.mean(costs))
            self._say('Test cost = ', self.last_test_cost)
            return self.last_test_cost


# ---------------------------------------------------------------------------------------------------
# Recipes (reference utils/trainer.py:392-658): WHAT each one wires, as data.  A recipe is a sequence of construction steps
# over `self.model` (the lazily-built @scope attributes make the ORDER part of the semantics: touching an attribute builds its
# sub-graph and creates its variables, so e.g. `back` must be touched before `saver` for the back-end to be restored).
#
#   ('load', who, key)      model = <who>.load(args[key], args)      who: 'sep' = the separator class, 'adapt' = Adapt
#   ('new', who)            model = <who>(**args)
#   ('args', {...})         args.update(...)
#   ('touch', 'a.b')        build the lazy attribute model.a.b
#   ('set', 'a.b', 'c.d')   model.a.b = model.c.d
#   ('call', 'm', ...)      model.m(...)          ('sep' as an argument stands for the separator class)
#   ('restore', key)        model.restore_model(args[key])
#   ('train_only_args',)    keep the trainable variables whose name contains one of args['train'] (--train, fine-tuning)
#   ('if', key, then, else) branch on args[key] is not None
# ---------------------------------------------------------------------------------------------------
_INIT_REST = ('call', 'initialize_non_init')
_SAVER = ('call', 'create_saver')
_FINISH = ('call', 'finish_construction')
_TB = ('call', 'tensorboard_init')
_OPT = ('touch', 'optimize')

WIRING = {
    # ---- inference (:392-463)
    'STFT_Separator_Enhanced_Inference': (
        ('load', 'sep', 'model_folder'), ('touch', 'separate'), ('touch', 'enhance'), ('set', 'output', 'postprocessing'),
        _SAVER, ('restore', 'model_folder'), _INIT_REST),
    'STFT_Separator_Inference': (
        ('load', 'sep', 'model_folder'), ('touch', 'separate'), ('set', 'output', 'postprocessing'),
        _SAVER, ('restore', 'model_folder'), _INIT_REST),
    'Front_Separator_Inference': (       # fine-tuned or not
        ('load', 'adapt', 'model_folder'), ('call', 'connect_front', 'sep'), ('set', 'sepNet.output', 'sepNet.separate'),
        ('set', 'output', 'back'), _SAVER, ('restore', 'model_folder'), _FINISH, _INIT_REST),
    'Front_Separator_Enhanced_Inference': (
        ('load', 'adapt', 'model_folder'), ('call', 'connect_front', 'sep'), ('set', 'sepNet.output', 'sepNet.enhance'),
        ('set', 'output', 'back'), _SAVER, ('restore', 'model_folder'), _INIT_REST),
    'Pretrained_Inference': (
        ('args', {'pretraining': True}), ('new', 'adapt'), ('set', 'output', 'back'), _SAVER, ('restore', 'model_folder')),
    # ---- training (:465-658)
    'STFT_Separator_Trainer': (
        ('if', 'model_folder',
         (('load', 'sep', 'model_folder'), _SAVER, ('restore', 'model_folder'), ('set', 'cost_model', 'cost'), _FINISH, _OPT, _TB,
          _INIT_REST),
         (('new', 'sep'), _TB, ('call', 'init_all'))),),
    'STFT_Separator_enhance_Trainer': (
        ('load', 'sep', 'model_folder'), _SAVER, ('restore', 'model_folder'), ('call', 'add_enhance_layer'), _TB, _INIT_REST),
    'STFT_Separator_FineTune_Trainer': (
        ('load', 'sep', 'model_folder'), ('touch', 'separate'), ('touch', 'enhance'), _SAVER, ('restore', 'model_folder'),
        ('touch', 'postprocessing'), ('set', 'cost_model', 'cost_finetuning'), _FINISH, ('train_only_args',), _OPT, _TB,
        _INIT_REST),
    'Adapt_Pretrainer': (('new', 'adapt'), _TB, ('call', 'init_all')),
    'Front_Separator_Trainer': (
        ('if', 'model_previous',
         (('load', 'adapt', 'model_previous'), ('call', 'connect_front', 'sep'), ('set', 'sepNet.output', 'sepNet.prediction'),
          ('set', 'cost_model', 'sepNet.cost'), ('touch', 'back'), _SAVER, ('restore', 'model_previous'), _FINISH,
          ('call', 'freeze_all_with', 'front/'), ('call', 'freeze_all_with', 'back/'), _OPT, _TB, _INIT_REST),
         (('load', 'adapt', 'model_folder'), ('call', 'connect_only_front_to_separator', 'sep'), _INIT_REST)),),
    'Front_Separator_Finetuning_Trainer': (
        ('load', 'adapt', 'model_folder'), ('call', 'connect_front', 'sep'), ('set', 'sepNet.output', 'sepNet.separate'),
        ('touch', 'back'), _SAVER, ('restore', 'model_folder'), ('set', 'cost_model', 'cost'), _FINISH,
        ('call', 'freeze_all_except', 'prediction', 'speaker_centroids'), _OPT, _TB, _INIT_REST),
    'Front_Separator_Enhance_Trainer': (
        ('load', 'adapt', 'model_folder'), ('call', 'connect_enhance_to_separator', 'sep'), _INIT_REST),
    'Front_Separator_Enhance_Finetuning_Trainer': (
        ('load', 'adapt', 'model_folder'), ('call', 'connect_front', 'sep'), ('set', 'sepNet.output', 'sepNet.enhance'),
        ('touch', 'back'), _SAVER, ('restore', 'model_folder'), ('set', 'cost_model', 'cost_finetuning'), _FINISH,
        ('train_only_args',), _OPT, _TB, _INIT_REST),
}


def _resolve(obj, path):
    for part in path.split('.'):
        obj = getattr(obj, part)
    return obj


def _wire(tr, steps):
    who = {'sep': lambda: tr.separator, 'adapt': lambda: Adapt}
    for st in steps:
        op = st[0]
        if op == 'load':
            tr.model = who[st[1]]().load(tr.args[st[2]], tr.args)
        elif op == 'new':
            tr.model = who[st[1]]()(**tr.args)
        elif op == 'args':
            tr.args.update(st[1])
        elif op == 'touch':
            _resolve(tr.model, st[1])
        elif op == 'set':
            owner, _, attr = st[1].rpartition('.')
            setattr(_resolve(tr.model, owner) if owner else tr.model, attr, _resolve(tr.model, st[2]))
        elif op == 'call':
            getattr(tr.model, st[1])(*[tr.separator if a == 'sep' else a for a in st[2:]])
        elif op == 'restore':
            tr.model.restore_model(tr.args[st[1]])
        elif op == 'train_only_args':
            wanted = tr.model.args['train']
            tr.model.trainable_variables = [v for v in tr.model.trainable_variables for p in wanted if p in v.ams_name]
        elif op == 'if':
            _wire(tr, st[2] if tr.args[st[1]] is not None else st[3])
        else:
            raise ValueError('unknown wiring step %r' % (st,))


def _make_recipe(name):
    """Recipe classes keep the reference's names and constructor signatures: Adapt_Pretrainer(**kwargs), every other one
    (separator, name, **kwargs)."""
    if name == 'Adapt_Pretrainer':
        def __init__(self, **kwargs):
            self.separator = None
            Trainer.__init__(self, trainer_type='pretraining', **kwargs)
    else:
        def __init__(self, separator, name, **kwargs):
            self.separator = separator
            Trainer.__init__(self, trainer_type=name, **kwargs)
    def build(self):
        _wire(self, WIRING[name])
        if name.endswith('_Inference') and not os.environ.get('AMS_NO_FREEZE'):
            self.model.freeze_weights()         # no optimizer in these recipes: weight-derived operands are kept across passes
    return type(name, (Trainer,), {'__init__': __init__, 'build': build,
                                   '__doc__': 'wiring: WIRING[%r]' % name})


for _name in WIRING:
    globals()[_name] = _make_recipe(_name)
del _name
