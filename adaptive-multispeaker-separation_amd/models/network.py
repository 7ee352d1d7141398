# -*- coding: utf-8 -*-
"""Host mirror of the reference's model-construction API (reference models/network.py).

Same classes, constructor kwargs, lazily-built attribute names and session-facing methods -- ``Network`` and
``Separator`` -- so ``utils/trainer.py`` recipes read like the reference's.  Attributes that were TF tensors are
``ams_hip.graph.Node`` objects evaluated once per ``train/valid_batch/...`` call; their bodies launch the HIP
kernels of libams_hip.so (no TF, no CPU fallback).
"""
import json
import os
import random

import numpy as np
import torch

import config
from ams_hip import functional as F
from ams_hip import ops as K
from ams_hip.graph import Node, Placeholder, Run, get_default_graph, scope
from ams_hip.optim import FlatOptimizer
from utils.ops import BLSTM, Conv1D, f_props

_ADJ = ['autumn', 'hidden', 'bitter', 'misty', 'silent', 'empty', 'dry', 'dark', 'summer', 'icy', 'quiet', 'white', 'cool',
        'spring', 'winter', 'patient', 'twilight', 'dawn', 'crimson', 'wispy', 'weathered', 'blue', 'billowing', 'broken']
_NOUN = ['waterfall', 'river', 'breeze', 'moon', 'rain', 'wind', 'sea', 'morning', 'snow', 'lake', 'sunset', 'pine', 'shadow',
         'leaf', 'dawn', 'glitter', 'forest', 'hill', 'cloud', 'meadow', 'sun', 'glade', 'bird', 'brook']


def haikunate():
    """Stand-in for haikunator.Haikunator().haikunate() (network.py:35): adjective-noun-4digits."""
    r = random.SystemRandom()
    return '%s-%s-%04d' % (r.choice(_ADJ), r.choice(_NOUN), r.randint(0, 9999))


class SummaryWriter(object):
    """JSON-lines stand-in for tf.summary.FileWriter (network.py:120-122); off the timed path."""

    def __init__(self, path):
        self.path = path
        self._f = None

    def add(self, step, values):
        if self._f is None:
            os.makedirs(self.path, exist_ok=True)
            self._f = open(os.path.join(self.path, 'events.jsonl'), 'a')
        self._f.write(json.dumps({'step': int(step), 'values': values}) + '\n')
        self._f.flush()


class Network(object):
    """docstring for Network"""

    def __init__(self, graph=None, *args, **kwargs):
        # Constant seed for uniform results (network.py:17-18)
        np.random.seed(42)

        if kwargs is not None and len(kwargs):
            self.folder = kwargs['type']
            self.S = kwargs['nb_speakers']
            if self.S > 4:
                # the permutation-invariant cost kernels (ams_pair_stats_*, ams_pair_combine_*: csrc/synth.hip) hold the S x S pair
                # table and the S! permutations of at most four speakers; say so at construction, not as AMS_E_INVALID_ARG in a step
                raise ValueError('--nb_speakers %d: the PIT / SDR cost kernels support at most 4 speakers' % self.S)
            self.args = kwargs
            self.learning_rate = kwargs['learning_rate']
            self.my_opt = kwargs['optimizer']
            self.decay_epoch = kwargs['decay_epoch']
            self.gradient_clip = kwargs['gradient_norm_clip']
        else:
            raise Exception('Keyword Arguments missing ! Please add the right arguments in input | check doc')

        self.dist = kwargs.get('dist', None)
        self.summaries_enabled = kwargs.get('summaries', True)

        if graph is None:
            # Run ID
            # one run folder for the whole job: with N ranks the id is chosen on rank 0 and broadcast (save() writes on rank 0
            # only, every rank restores from that folder at the end of Trainer.train)
            self.runID = kwargs.get('run_id') or haikunate()
            if self.dist is not None and getattr(self.dist, 'enabled', False) and not kwargs.get('run_id'):
                self.runID = self.dist.broadcast_object(self.runID)
            print('ID : {}'.format(self.runID))
            g = get_default_graph()
            with g.variable_scope('inputs'):
                self.training = Placeholder('is_training')
                if not kwargs['pipeline']:
                    # fed through feed_dict (network.py:44-63)
                    self.x_non_mix = Placeholder('non_mix_input')      # [B, S, L]
                    self.x_mix = Placeholder('mix_input')              # [B, L]
                    self.I = Placeholder('indicies')                   # [B, S]
                else:
                    # tensors produced by the input pipeline (network.py:65-85)
                    nm, mx, ind = kwargs['non_mix'], kwargs['mix'], kwargs['ind']
                    self.x_non_mix = Node('non_mix_input', lambda run: nm.value(run))
                    self.x_mix = Node('mix_input', lambda run: mx.value(run))
                    self.I = Node('indicies', lambda run: ind.value(run))

    # --------------------------------------------------------------- bookkeeping
    def _dir(self):
        return os.path.join(config.log_dir, self.folder, self.runID)

    def tensorboard_init(self):
        self.create_saver()
        g = get_default_graph()
        train_keys, valid_keys, test_keys = [], [], []
        for name in g.summaries:                                        # network.py:98-107
            if not ('input' in name or 'output' in name):
                train_keys.append(name)
            else:
                valid_keys.append(name)
            if 'SDR_improvement' in name:
                valid_keys.append(name)
                test_keys.append(name)
            if 'audio' in name or 'stft' in name or 'mask' in name:
                test_keys.append(name)
        self.merged_train = train_keys
        self.merged_valid = valid_keys if len(valid_keys) else None
        self.merged_test = test_keys if len(test_keys) else None
        self.train_writer = SummaryWriter(os.path.join(self._dir(), 'train'))
        self.valid_writer = SummaryWriter(os.path.join(self._dir(), 'valid'))
        self.test_writer = SummaryWriter(os.path.join(self._dir(), 'test'))
        # Save arguments (network.py:124-129)
        if self.dist is None or self.dist.rank == 0:
            os.makedirs(self._dir(), exist_ok=True)
            with open(os.path.join(self._dir(), 'params'), 'w') as f:
                for k in ('mix', 'non_mix', 'ind'):
                    self.args.pop(k, None)
                json.dump({k: v for k, v in self.args.items() if _jsonable(v)}, f)

    def create_saver(self, subset=None):
        g = get_default_graph()
        self.saver = list(g.variables.keys()) if subset is None else [v.ams_name for v in subset]

    def _latest_checkpoint(self, path):
        marker = os.path.join(path, 'checkpoint')
        if os.path.exists(marker):
            with open(marker) as f:
                name = json.load(f)['model_checkpoint_path']
            return os.path.join(path, name)
        cands = sorted([p for p in os.listdir(path) if p.startswith('model-') and p.endswith('.npz')],
                       key=lambda p: int(p[6:-4]))
        if not cands:
            raise IOError('no checkpoint in %s' % path)
        return os.path.join(path, cands[-1])

    # Restore last checkpoint of the current graph using the total path (network.py:139-140)
    def restore_model(self, path):
        g = get_default_graph()
        # a folder written by the REFERENCE (tf.train.Saver: text `checkpoint` file + model-N.index/.data-*) is read directly
        from ams_hip import tf_checkpoint
        marker = os.path.join(path, 'checkpoint')
        is_tf = os.path.exists(marker) and open(marker).read(22).startswith('model_checkpoint_path')
        if is_tf:
            prefix = tf_checkpoint.latest_checkpoint(path)
            bundle = tf_checkpoint.read_bundle(prefix, names=set(self.saver))
            for name in self.saver:
                if name not in bundle:
                    # the BLSTM / dense / speaker-vector names are inferred from the reference's scopes, never
                    # observed in a TensorFlow-written file: say what the bundle does hold, so a renamed variable is one edit away
                    have = sorted(tf_checkpoint.list_bundle(prefix)[1])
                    raise KeyError('variable %s not found in TensorFlow checkpoint %s; the bundle holds: %s' % (name, prefix, ', '.join(have)))
                v = g.variables[name]
                arr = bundle[name]
                if tuple(arr.shape) != tuple(v.shape):
                    # the reference's Conv1D kernels are 3-D [1, Din, Dout] (utils/ops.py:486-492); here they are [Din, Dout]
                    squeezed = tuple(d for d in arr.shape if d != 1)
                    if squeezed != tuple(d for d in v.shape if d != 1):
                        raise ValueError('variable %s: checkpoint shape %s, graph shape %s' % (name, arr.shape, tuple(v.shape)))
                    arr = arr.reshape(tuple(v.shape))
                v.data.copy_(torch.from_numpy(arr.astype(np.float32)).to(v.device))
                g.initialized.add(name)
            self._weights_written()
            return
        data = np.load(self._latest_checkpoint(path))
        for name in self.saver:
            key = name.replace('/', '.')
            if key not in data.files:
                raise KeyError('variable %s not found in checkpoint %s' % (name, path))
            v = g.variables[name]
            v.data.copy_(torch.from_numpy(data[key]).to(v.device))
            g.initialized.add(name)
        self._weights_written()

    @staticmethod
    def _weights_written():
        """Variables were written by something other than an optimizer kernel: what captured steps keep across replays instead of
        re-deriving it from the weights (the weights' operand bound, a frozen front's filter) must be refreshed."""
        g = get_default_graph()
        g.weights_epoch = getattr(g, 'weights_epoch', 0) + 1
        if torch.cuda.is_available():
            K.param_bounds_dirty()
        for v in g.variables.values():                 # (freeze_weights: what was derived from the old values)
            K.drop_frozen_derivatives(v)

    def freeze_weights(self):
        """Inference recipes (no optimizer is ever built): nothing writes the variables between passes, so what a pass derives from
        them alone -- operand bounds of the fp16x3 products, the gathered [D, 8H] projection kernels, concatenated biases -- is kept
        across passes (ams_hip/ops.py::_frozen) instead of being re-derived in each: ~45 us of launches per BLSTM layer and pass in
        an eager inference step, and the recurrent product of the rings gets a bound, i.e. runs as fp16x3 as in training.
        restore_model() / anything that calls _weights_written() drops the derived tensors; code that writes `.data` of a variable
        behind the model's back must call _weights_written() itself (as it must for captured steps)."""
        for v in get_default_graph().variables.values():
            K.drop_frozen_derivatives(v)
            v._ams_frozen = True

    def restore_last_checkpoint(self):
        self.restore_model(self._dir())

    def init_all(self):
        g = get_default_graph()
        g.initialized.update(g.variables.keys())      # variables are materialised with their initial value

    def non_initialized_variables(self):
        g = get_default_graph()
        return [n for n in g.variables if n not in g.initialized]

    def initialize_non_init(self):
        names = self.non_initialized_variables()
        print('not init: ', names)
        get_default_graph().initialized.update(names)

    @scope
    def optimize(self):
        print('Train the following variables :', [v.ams_name for v in self.trainable_variables])
        g = get_default_graph()
        for v in g.global_variables():
            v.requires_grad_(False)
        opt = FlatOptimizer(self.trainable_variables, self.my_opt, self.learning_rate, self.decay_epoch,
                            self.gradient_clip, self.dist)
        F.OVERLAP.enabled = (bool(self.args.get('overlap_weight_grads', True)) and torch.cuda.is_available()
                             and os.environ.get('AMS_OVERLAP', '1') != '0')
        F.OVERLAP.on_ready = opt.bucket_ready if getattr(opt, 'overlap', False) else None
        # Variables this recipe does NOT train (a restored separator under an enhance stack, a restored front end) are written by nobody
        # between passes -- the optimizer owns only the trainable ones, and restore_model() / _weights_written() drop what was derived:
        # what a pass derives from them alone (operand bounds, pre-split images, gathered kernels) is kept across passes as in the
        # inference recipes (freeze_weights): 13 absmax + fill pairs and 4 image cuts per STFT_L41_enhance step otherwise
        if os.environ.get('AMS_FREEZE_UNTRAINED', '1') != '0':
            trained = set(id(v) for v in self.trainable_variables)
            for v in g.variables.values():
                if id(v) not in trained:
                    K.drop_frozen_derivatives(v)
                    v._ams_frozen = True
        self.optimizer = opt
        self.increment_epoch = opt.increment_epoch
        g.summaries['optimize/learning_rate'] = Node('learning_rate', lambda run: opt.learning_rate())
        return opt

    def sdr_improvement(self, s_target, s_approx, with_perm=False):
        """network.py:196-221 -- built by models that own a waveform output (Adapt.back, postprocessing)."""
        from ams_hip import losses_host
        return losses_host.sdr_improvement(self, s_target, s_approx, with_perm)

    # --------------------------------------------------------------- checkpoints
    def save(self, step):
        path = os.path.join(self._dir(), 'model')
        if self.dist is None or self.dist.rank == 0:
            g = get_default_graph()
            os.makedirs(self._dir(), exist_ok=True)
            arrays = {n.replace('/', '.'): g.variables[n].detach().cpu().numpy() for n in self.saver}
            fname = 'model-%d.npz' % step
            np.savez(os.path.join(self._dir(), fname), **arrays)
            with open(os.path.join(self._dir(), 'checkpoint'), 'w') as f:
                json.dump({'model_checkpoint_path': fname}, f)
        return path

    # --------------------------------------------------------------- session-facing methods
    def _feeds(self, feed_dict, training, new_pass=True):
        feeds = dict(feed_dict)
        feeds[self.training] = training
        return Run(feeds, training, new_pass)

    def _summaries(self, run, keys):
        g = get_default_graph()
        out = {}
        for k in keys or []:
            v = g.summaries[k].value(run)
            out[k] = float(v) if not torch.is_tensor(v) else float(v.detach().reshape(-1)[0].item())
        return out

    # ---- hipGraph replay of forward+backward (the launch-bound inner loops: 80 recurrent steps x 6 cells x 2) ----
    def _fetch_inputs(self, feed_dict):
        """One batch from the input pipeline (a probe Run that only evaluates the input nodes)."""
        # the probe only fetches the input nodes: it must NOT begin a pass (ops.pass_begin clears the ring arena and rewrites the shared
        # weight bound on the side stream, un-joined, while the replayed graph owns both addresses) nor count as one
        probe = self._feeds(feed_dict, True, new_pass=False)
        return [n.value(probe) for n in (self.x_mix, self.x_non_mix, self.I)]

    @staticmethod
    def _back_to_back(a, b):
        return (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.device == b.device and
                a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and
                a.data_ptr() + a.numel() * a.element_size() == b.data_ptr())

    @staticmethod
    def _static_like(ins):
        """Static input buffers of a captured step: x_mix [B,L] and x_non_mix [B,S,L] back to back in one buffer (Adapt's
        concat([x_mix, x_non_mix rows]) is then a view), the rest cloned."""
        xm, xn = ins[0], ins[1]
        if xm.dtype == xn.dtype and xm.dim() == 2 and xn.dim() == 3 and xm.shape[-1] == xn.shape[-1]:
            flat = torch.empty(xm.numel() + xn.numel(), dtype=xm.dtype, device=xm.device)
            sm, sn = flat[:xm.numel()].view(xm.shape), flat[xm.numel():].view(xn.shape)
            sm.copy_(xm)
            sn.copy_(xn)
            return [sm, sn] + [t.clone() for t in ins[2:]]
        return [t.clone() for t in ins]

    def _stage(self, static, ins):
        """This batch into the static buffers of the captured step.  x_mix and x_non_mix adjacent on both sides (_static_like; the
        synthetic pool stores them so): ONE launch moves them, the small third input (speaker indices) with them, and leaves
        max |waveform| -- the front product's operand bound -- on the way (K.stage_inputs); anything else: plain copies."""
        pairs = list(zip(static, ins))
        if (self._back_to_back(static[0], static[1]) and self._back_to_back(ins[0], ins[1]) and ins[0].is_cuda
                and ins[0].dtype == torch.float32 and ins[0].data_ptr() % 16 == 0 and static[0].data_ptr() % 16 == 0):
            n = ins[0].numel() + ins[1].numel()
            src2 = dst2 = None
            rest = pairs[2:]
            if len(rest) == 1 and rest[0][1].is_contiguous() and rest[0][0].is_contiguous() and rest[0][1].numel() * rest[0][1].element_size() <= 65536:
                dst2, src2 = rest[0]
                rest = []
            am = K.stage_inputs(torch.as_strided(ins[0], (n,), (1,)), torch.as_strided(static[0], (n,), (1,)), src2, dst2)
            static[0]._ams_x_amax = am              # models/adapt.py::Adapt._x tags the concatenated view with it
            pairs = rest
            for dst, src in pairs:
                dst.copy_(src)
            return
        for dst, src in pairs:
            dst.copy_(src)
        am = getattr(static[0], '_ams_x_amax', None)
        if am is not None and static[0].is_cuda:
            # an earlier batch went through the fused launch and the captured step holds the address of its bound: this batch came by
            # plain copies (not back to back / not aligned), so the bound is measured over what was just copied -- never left stale
            # (a louder batch under a stale max |x| overflows fp16 in the front product: ADVICE r05)
            flat = static[0]
            if self._back_to_back(static[0], static[1]):
                flat = torch.as_strided(static[0], (static[0].numel() + static[1].numel(),), (1,))
                K.absmax(flat, out=am)
            else:
                K.absmax(static[0].reshape(-1), out=am)
                K.absmax(static[1].reshape(-1), out=(tmp := torch.empty_like(am)))
                torch.maximum(am, tmp, out=am)

    def _train_graphed(self, feed_dict, step):
        """Capture zero_grad + forward + backward once (after 2 eager steps) and replay it; inputs are copied into
        static buffers, the optimizer (per-step lr_t, all-reduce) stays outside the graph.
        (Round 4 built a two-graph form that computed a FROZEN front end one batch ahead, beside the previous step's forward rings:
        measured 0 ... -0.5 %, not kept -- profiles/r04_d_front_ahead_ab.txt, commits 43a78ea..dc02139.)"""
        st = self.__dict__.setdefault('_cg_state', {'n': 0, 'graph': None,
                                                    'stream': torch.cuda.Stream(priority=int(os.environ.get('AMS_MAIN_PRIORITY', '0')))})
        ins = self._fetch_inputs(feed_dict)
        opt = self.optimize
        side = st['stream']
        epoch = getattr(get_default_graph(), 'weights_epoch', 0)
        if st['graph'] is not None and st.get('epoch') != epoch:
            # somebody other than the optimizer kernel wrote weights since the capture (restore_model): what the captured step took as
            # constants -- a frozen front's filter and its bound (models/adapt.py) -- is stale.  Capture again behind two eager
            # steps, which re-derive them.
            st['graph'], st['n'] = None, 0
        if st['graph'] is None:
            st['n'] += 1
            if st['n'] <= 2:
                # eager warm-up ON THE CAPTURE STREAM, so autograd's AccumulateGrad nodes are bound to it
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    run = self._feeds(feed_dict, True)
                    for node, t in zip((self.x_mix, self.x_non_mix, self.I), ins):
                        run.cache[id(node)] = t
                    opt.zero_grad(defer=True)
                    cost = self.cost_model.value(run)
                    self._backward(cost)
                    F.OVERLAP.join()
                torch.cuda.current_stream().wait_stream(side)
                opt.step()
                self.last_run = run
                return cost.detach().reshape(-1)[0]
            st['static'] = self._static_like(ins)
            self._stage(st['static'], ins)          # (also attaches the waveforms' bound to the static buffer before the capture reads it)

            def capture(presplit):
                run = self._feeds(feed_dict, True)
                for node, t in zip((self.x_mix, self.x_non_mix, self.I), st['static']):
                    run.cache[id(node)] = t
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                # thread_local: a collective library's watchdog thread (RCCL at N > 1) must not be able to invalidate this capture
                with K.presplit(presplit), torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                    opt.zero_grad(defer=True)
                    cost = self.cost_model.value(run)
                    self._backward(cost)
                    F.OVERLAP.join()
                return g, cost, run
            # Which form of the forward products this step replays (K.PS_AUTOTUNE): decided once per model by measurement, see below
            choice = self.__dict__.get('_ps_choice')
            n0 = K.PS_LAUNCHES[0]
            first = K.PRESPLIT if choice is None else (choice and K.PRESPLIT)
            st['graph'], st['cost'], st['run'] = capture(first)
            st['epoch'], st['tune'] = epoch, None
            st['twin'] = capture if (K.PS_AUTOTUNE and (K.PS_LAUNCHES[0] > n0 or choice is not None)) else None      # (closes over this capture's static buffers)
            if choice is None and K.PS_AUTOTUNE and first and K.PS_LAUNCHES[0] > n0:
                # the captured step holds pre-split products: capture its twin without them and let the next steps decide.  Both are
                # the same arithmetic to the last bits (tests/test_gpu_gemm_ps.py); every tuning step is a real training step.
                st['tune'] = {'variants': [(st['graph'], st['cost'], st['run']), capture(False)], 'k': 0, 'ev': ([], [])}
        tune = st.get('tune')
        if (tune is None and self._PS_RETUNE_EVERY > 0 and K.PS_AUTOTUNE and self.__dict__.get('_ps_choice') is not None
                and st.get('since_tune', 0) >= self._PS_RETUNE_EVERY and st.get('twin') is not None):
            # a long run asks the question again (what a board's governor allows moves with its temperature): the form that lost is
            # captured anew and the two are measured as at the start
            keep = (st['graph'], st['cost'], st['run'])
            other = st['twin'](not self._ps_choice)
            st['tune'] = tune = {'variants': [keep, other] if self._ps_choice else [other, keep], 'k': 0, 'ev': ([], [])}
        st['since_tune'] = 0 if tune is not None else st.get('since_tune', 0) + 1
        if tune is not None:
            v = (tune['k'] // self._PS_TUNE_BLOCK) % 2
            st['graph'], st['cost'], st['run'] = tune['variants'][v]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self._stage(st['static'], ins)
        for hook in get_default_graph().pre_replay_hooks:      # host-drawn inputs of captured kernels (k-means seeds)
            hook()
        st['graph'].replay()
        opt.step()
        self.last_run = st['run']
        cost = st['cost'].detach().reshape(-1)[0]
        if tune is not None:
            e1.record()
            tune['ev'][v].append((tune['k'] % self._PS_TUNE_BLOCK, e0, e1))
            tune['k'] += 1
            if tune['k'] >= 4 * self._PS_TUNE_BLOCK:
                self._ps_finish_tuning(st)
        return cost

    # steps per block; blocks alternate pre-split / in-product twice; only the last third of a block is counted: the clock governor takes
    # tens of milliseconds to settle after the form changes (blocks of 8 steps measured both forms at the slower form's clock)
    _PS_TUNE_BLOCK = int(os.environ.get('AMS_PS_TUNE_BLOCK', '48'))
    _PS_RETUNE_EVERY = int(os.environ.get('AMS_PS_RETUNE_EVERY', '50000'))       # replays between two measurements (0 = decide once)

    def _ps_finish_tuning(self, st):
        """Median step time (replay + optimizer, device events) of the two captured forms over the counted tuning steps; the faster
        form stays, the other graph is dropped.  The decision is kept on the model (a re-capture -- an audit, a restore -- reuses it)
        and published in K.PS_TUNED for reports."""
        tune = st['tune']
        torch.cuda.synchronize()
        med = []
        for v in (0, 1):
            t = sorted(a.elapsed_time(b) for i, a, b in tune['ev'][v] if i >= 2 * self._PS_TUNE_BLOCK // 3)
            med.append(t[len(t) // 2])
        keep = 0 if med[0] <= med[1] else 1
        st['graph'], st['cost'], st['run'] = tune['variants'][keep]
        st['tune'] = None
        self._ps_choice = (keep == 0)
        K.PS_TUNED.update(presplit=self._ps_choice, ms_presplit=round(med[0], 4), ms_in_product=round(med[1], 4), decisions=K.PS_TUNED.get('decisions', 0) + 1,
                          steps_counted=len([1 for i, a, b in tune['ev'][0] if i >= 2 * self._PS_TUNE_BLOCK // 3]))

    def _backward(self, cost):
        """d cost[0]: a 1-element cost is seeded with a cached ones tensor (no select / fill launches on the way back)."""
        K.flush_deferred_zero()          # zero_grad(defer=True): zeroed by pass_begin() on the side stream, or here if no pass began
        if cost.is_cuda:
            K.await_pass_side()          # gradients written on this stream come after that memset
        c = cost.reshape(-1)
        if c.numel() != 1:
            c[0].backward()
            return
        seed = self.__dict__.get('_seed')
        if seed is None or seed.device != c.device:
            seed = self.__dict__['_seed'] = torch.ones(1, dtype=c.dtype, device=c.device)
        c.backward(seed)

    def train(self, feed_dict, step):
        if self.args.get('hip_graph'):
            c = self._train_graphed(feed_dict, step)
            if c is not None:
                return c
        run = self._feeds(feed_dict, True)
        opt = self.optimize
        opt.zero_grad(defer=True)
        cost = self.cost_model.value(run)
        self._backward(cost)
        F.OVERLAP.join()
        opt.step()
        if self.summaries_enabled and getattr(self, 'merged_train', None):
            with torch.no_grad():
                self.train_writer.add(step, self._summaries(run, self.merged_train))
        self.last_run = run
        return cost.detach().reshape(-1)[0]

    def train_audited(self, feed_dict, step):
        """One training step, EAGER, with the fp16x3 range audit on (K.F16_AUDIT): every product launched with operand bounds also
        measures how much of each operand lies below bound * 2^-17.  A product class whose lost bits exceed the f32 level is sent back
        to bf16x6 from then on; a captured step is dropped so that the next call re-captures without it.  Returns (cost, newly denied)."""
        K.F16_AUDIT.begin()
        hg, self.args['hip_graph'] = self.args.get('hip_graph'), False
        try:
            c = self.train(feed_dict, step)
        finally:
            self.args['hip_graph'] = hg
            new = K.F16_AUDIT.finish()
        if new:
            self.__dict__.pop('_cg_state', None)
        return c, new

    # ---- a recurrence ring that could not get all its workgroups resident gives up a bounded wait and flags it in the device's
    # sticky error word (csrc/lstm_ring.hip, K.ring_error_word): the batch is then REPEATED on the per-step recurrence kernels,
    # which need no co-residency -- safe rather than loud.  `ring_fallbacks` counts how often that happened.
    ring_fallbacks = 0

    def _inputs_of(self, run):
        return [(n, n.value(run)) for n in (self.x_mix, self.x_non_mix, self.I)]

    def _eval_guarded(self, feed_dict, fn, training=False):
        """fn(run) without gradients; the same batch again on the per-step kernels when a ring launch gave up (one host sync: every
        caller reads its results on the host anyway)."""
        # a word that is ALREADY set belongs to an earlier (training) launch whose owner has yet to handle it (Trainer.train checks
        # right after the step): it is not this evaluation's to clear -- the batch is evaluated on the per-step kernels and the word
        # stays up
        pre = K.LSTM_RING != '0' and self._ring_error_any()
        run = self._feeds(feed_dict, training)
        if not pre:
            with torch.no_grad():
                out = fn(run)
        # (pre: the word is up already -- a ring evaluation would be run, thrown away and leave nothing new to learn: straight to the
        # per-step kernels, one collective per batch instead of two)
        if K.LSTM_RING != '0' and (pre or self._ring_error_any()):
            ins = self._inputs_of(run)
            if not pre:
                K.ring_errors_clear()
            old, K.LSTM_RING = K.LSTM_RING, '0'
            try:
                run = self._feeds(feed_dict, training)
                for n, t in ins:
                    run.cache[id(n)] = t
                with torch.no_grad():
                    out = fn(run)
            finally:
                K.LSTM_RING = old
            Network.ring_fallbacks += 1
        return out

    def _ring_error_any(self):
        """Did a ring launch of the evaluation just run give up -- on ANY rank?  The decision to repeat a batch must be the same on
        every rank: the repeat may issue collectives (SparseKL all-reduces p_hat) its peers would otherwise not match."""
        d = self.dist
        if d is not None and getattr(d, 'enabled', False) and torch.cuda.is_available():
            d.all_reduce_max(K.ring_error_word())
        return K.ring_error_pending()

    def retrain_last(self, step):
        """Repeat the LAST training step on the per-step recurrence kernels.  Call when K.ring_error_pending() after train():
        the fused optimizer has skipped that step's update (the error word is its guard), so weights and slots are those of before
        the step; this runs the same batch eagerly with the ring off and applies the update.  Returns the cost."""
        run0 = self.last_run
        ins = self._inputs_of(run0)
        opt = self.optimize
        K.ring_errors_clear()
        opt.undo_counters()
        old, K.LSTM_RING = K.LSTM_RING, '0'
        try:
            run = Run(dict(run0.feeds), True)
            for n, t in ins:
                run.cache[id(n)] = t
            opt.zero_grad()
            cost = self.cost_model.value(run)
            self._backward(cost)
            F.OVERLAP.join()
            opt.step()
        finally:
            K.LSTM_RING = old
        Network.ring_fallbacks += 1
        self.last_run = run
        return cost.detach().reshape(-1)[0]

    def infer(self, feed_dict, step):
        return self._eval_guarded(feed_dict, lambda run: [self.x_mix.value(run), self.x_non_mix.value(run), self.output.value(run)])

    def improvement(self, feed_dict, step):
        return self._eval_guarded(feed_dict, lambda run: [self.x_mix.value(run), self.x_non_mix.value(run), self.sdr_imp.value(run)])

    def valid_batch(self, feed_dict, step):
        def fn(run):
            cost = self.cost_model.value(run)
            return run, float(cost.reshape(-1)[0].item())
        run, c = self._eval_guarded(feed_dict, fn)
        if getattr(self, 'merged_valid', None) and self.summaries_enabled:
            with torch.no_grad():
                self.valid_writer.add(step, self._summaries(run, self.merged_valid))
        return c

    def get_embeddings(self, feed_dict):
        return self._eval_guarded(feed_dict, lambda run: self.prediction.value(run))

    def test_batch(self, feed_dict):
        return self._eval_guarded(feed_dict, lambda run: float(self.cost_model.value(run).reshape(-1)[0].item()))

    def test(self, feed_dict):
        run = self._feeds(feed_dict, True)
        with torch.no_grad():
            return self.y.value(run)

    def add_valid_summary(self, val, step):
        self.valid_writer.add(step, {'Valid Cost': float(val)})

    def freeze_all_with(self, prefix):
        self.trainable_variables = [v for v in self.trainable_variables if prefix not in v.ams_name]

    def freeze_all_except(self, *prefix):
        to_train = []
        for var in self.trainable_variables:
            for p in prefix:
                if p in var.ams_name:
                    to_train.append(var)
                    break
        self.trainable_variables = to_train

    @classmethod
    def load(cls, path, modified_args):
        # Load parameters used for the desired model to load (network.py:291-306)
        params_path = os.path.join(path, 'params')
        with open(params_path) as f:
            args = json.load(f)
            keys_to_update = ['learning_rate', 'epochs', 'batch_size', 'chunk_size', 'nb_speakers',
                              'regularization', 'overlap_coef', 'loss', 'beta', 'model_folder', 'type', 'pretraining',
                              'with_silence', 'end_assign', 'beta_kmeans', 'nb_tries', 'nb_steps', 'threshold', 'optimizer',
                              'men', 'women', 'recurrent_dropout', 'recurrent_dropout_enhance',
                              # this build's own switches follow the CURRENT command line, not the loaded model's
                              'hip_graph', 'no_summaries', 'summaries', 'synthetic_batches', 'synthetic_pool', 'run_id',
                              'kmeans_seeding', 'f16_audit_every', 'dist']
            to_modify = {key: modified_args[key] for key in keys_to_update if key in modified_args.keys()}
            to_modify.update({key: val for key, val in modified_args.items() if key not in args.keys()})
        args.update(to_modify)
        print("LOADED = ", {k: v for k, v in args.items() if _jsonable(v)})
        return cls(**args)

    def finish_construction(self):
        self.trainable_variables = get_default_graph().global_variables()


def _jsonable(v):
    try:
        json.dumps(v)
        return True
    except (TypeError, ValueError):
        return False


from models.Kmeans_2 import KMeans  # noqa: E402  (reference import order, network.py:311)


class Separator(Network):

    def __init__(self, plugged=False, *args, **kwargs):
        super(Separator, self).__init__(plugged, *args, **kwargs)

        self.num_speakers = kwargs['tot_speakers']
        self.layer_size = kwargs['layer_size']
        self.embedding_size = kwargs['embedding_size']
        self.normalize = kwargs['no_normalize']
        self.nb_layers = kwargs['nb_layers']
        self.a = kwargs['mask_a']
        self.b = kwargs['mask_b']
        self.rdropout = kwargs['recurrent_dropout']

        # Preproc
        self.normalize_input = kwargs['normalize_separator']
        self.abs_input = kwargs['abs_input']
        self.pre_func = kwargs['pre_func']
        self.silent_threshold = kwargs['silence_mask_db']

        # Loss Parameters
        self.loss_with_silence = kwargs['silence_loss']
        self.threshold_silence_loss = kwargs['threshold_silence_loss']
        self.function_mask = kwargs['function_mask']

        # Kmeans Parameters
        self.beta = kwargs['beta_kmeans']
        self.threshold = kwargs['threshold']
        self.with_silence = kwargs['with_silence']
        self.nb_tries = kwargs['nb_tries']
        self.nb_steps = kwargs['nb_steps']

        # Negative Sampling for L41
        self.sampling = kwargs['sampling']
        self.ns_rate = kwargs['ns_rate']
        self.ns_method = kwargs['ns_method']

        self.add_dilated = kwargs['add_dilated']
        if self.add_dilated:
            raise NotImplementedError('--add_dilated (experimental dilated front, network.py:527-551) is out of scope')

        self.graph = get_default_graph()
        self.plugged = plugged
        # If the Separator is not independant but using a front layer (network.py:357-400)
        if self.plugged:
            self.F = kwargs['filters']
            g = self.graph
            self.training = g.get_tensor_by_name('inputs/is_training:0')
            front = g.get_tensor_by_name('front/output:0')
            non_mix_in = g.get_tensor_by_name('inputs/non_mix_input:0')
            self.x_non_mix = non_mix_in
            self.x_mix = g.get_tensor_by_name('inputs/mix_input:0')
            S, Fq = self.S, self.F

            with g.variable_scope('split_front'):
                def _x(run):
                    B = non_mix_in.value(run).shape[0]
                    return front.value(run)[:B]                          # [B, T, N] signed mixture representation
                self.X = Node('X', _x)
                x_split = self.X
                self.X_input = Node('X_input', lambda run: x_split.value(run))     # tf.identity(self.X), network.py:371

                def _xnm(run):
                    B = non_mix_in.value(run).shape[0]
                    return front.value(run)[B:]                          # rows (b,s): [B*S, T, N]
                self.X_non_mix_rows = Node('X_non_mix_rows', _xnm)
                self.X_non_mix = Node('X_non_mix', lambda run: self.X_non_mix_rows.value(run).reshape(
                    -1, S, self.X_non_mix_rows.value(run).shape[1], Fq).permute(0, 2, 3, 1))

            with g.variable_scope('create_masks'):
                def _y(run):
                    rows = self.X_non_mix_rows.value(run)
                    B = rows.shape[0] // S
                    # a DPCL training step feeds these labels to the fused loss: count them in the pass that writes them
                    weighted = self.function_mask in ('linear', 'sqrt', 'square') or self.loss_with_silence
                    E = getattr(self, 'count_labels_for', None) if (run.training and not weighted) else None
                    y = K.make_masks(rows, B, S, self.a, self.b, True, dpcl_E=E)  # [B, TF, S]
                    return self._weight_masks(y, run).reshape(B, rows.shape[1], Fq, S)
                self.y = Node('y', _y)

            self.I = g.get_tensor_by_name('inputs/indicies:0')
        else:
            # STFT hyperparams
            self.window_size = kwargs['window_size']
            self.hop_size = kwargs['hop_size']
            self.F = kwargs['window_size'] // 2 + 1

    def _weight_masks(self, y, run):
        """network.py:381-396: magnitude-weighted masks / silence loss mask (off in the shipped launchers)."""
        if self.function_mask in ('linear', 'sqrt', 'square') or self.loss_with_silence:
            X = self.X.value(run)
            mode = self.function_mask if self.function_mask in ('linear', 'sqrt', 'square') else None
            y = K.weight_masks(X.contiguous(), y.contiguous(), mode,
                               self.threshold_silence_loss if self.loss_with_silence else None)
        return y

    def init_separator(self):
        if self.plugged:
            if self.abs_input:
                src = self.X
                self.X = Node('abs_input', lambda run: K.row_transform(src.value(run).contiguous(), pre='abs'), register=False)
            if self.normalize_input == '01':
                self.normalization01
            elif self.normalize_input == 'meanstd':
                self.normalization_mean_std
            self.prediction
        else:
            # STFT
            self.preprocessing
            if self.pre_func in ('sqrt', 'log'):
                src, fn = self.X, self.pre_func
                self.X = Node('pre_func', lambda run: K.row_transform(src.value(run).contiguous(), pre=fn), register=False)
            if self.normalize_input == '01':
                self.normalization01
            elif self.normalize_input == 'meanstd':
                self.normalization_mean_std
            if self.silent_threshold > 0:
                src, thr = self.X, self.silent_threshold

                def _sil(run):
                    return K.row_transform(src.value(run).contiguous(), norm='silent', thr=thr / 20.)
                self.X = Node('silent_mask', _sil, register=False)
            self.prediction
            if self.args['model_folder'] is None:
                self.cost_model = self.cost
                self.finish_construction()
                self.optimize

    def add_enhance_layer(self):
        self.separate
        self.enhance
        self.cost_model = self.enhance_cost
        self.finish_construction()
        self.freeze_all_except('enhance')
        self.optimize

    def add_finetuning(self):
        self.separate
        self.enhance
        self.postprocessing
        self.cost_finetuning
        self.cost_model = self.cost_finetuning
        self.finish_construction()
        to_train = []
        for var in self.trainable_variables:
            for p in self.args['train']:
                if p in var.ams_name:
                    to_train.append(var)
        self.trainable_variables = to_train
        self.optimize

    @scope
    def preprocessing(self):
        from ams_hip import stft_host
        return stft_host.build_preprocessing(self)

    @scope
    def normalization01(self):
        src = self.X

        def _n(run):
            return K.row_transform(src.value(run).contiguous(), norm='01')
        self.X = Node('X01', _n)
        return self.X

    @scope
    def normalization_mean_std(self):
        src = self.X

        def _n(run):
            return K.row_transform(src.value(run).contiguous(), norm='meanstd')
        self.X = Node('Xms', _n)
        return self.X

    @scope
    def prediction(self):
        pass

    @scope
    def separate(self):
        from ams_hip import separate_host
        return separate_host.build_separate(self, KMeans)

    @scope
    def postprocessing(self):
        from ams_hip import stft_host
        return stft_host.build_postprocessing(self)

    @scope
    def enhance(self):
        from ams_hip import separate_host
        return separate_host.build_enhance(self, BLSTM, Conv1D, f_props)

    @scope
    def enhance_cost(self):
        from ams_hip import separate_host
        return separate_host.build_enhance_cost(self)

    @scope
    def cost_finetuning(self):
        from ams_hip import separate_host
        return separate_host.build_cost_finetuning(self, self.postprocessing)
