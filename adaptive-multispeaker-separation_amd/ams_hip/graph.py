"""Minimal deferred-evaluation layer that gives the host mirror the reference's construction semantics.

The reference is a TF-1.x static graph: accessing a ``@scope`` property (utils/ops.py:34-44) builds a
sub-graph once inside ``tf.variable_scope(<method name>)``; ``sess.run`` later executes it per batch, and
plugged separators find tensors BY NAME ('front/output:0', models/network.py:362-366,400).  Here:

  * ``Graph`` holds named variables (device tensors, creation order = ``tf.global_variables()`` order) and
    named nodes; a default-graph stack mirrors ``tf.Graph().as_default()``;
  * a ``Node`` is a named thunk evaluated once per ``Run`` (= one ``sess.run``); evaluating it launches the
    HIP kernels through ams_hip.functional (autograd-aware);
  * ``scope`` is the cached lazy property + variable-name scoping of utils/ops.py:34-44.
"""
import contextlib
import functools
from collections import OrderedDict

import torch

_default_stack = []


class Graph(object):
    def __init__(self, device=None):
        self.variables = OrderedDict()          # name -> tensor (requires_grad toggled by optimize)
        self.initialized = set()                # names restored from a checkpoint or initialised
        self.nodes = {}
        self.scope_stack = []
        self.summaries = OrderedDict()          # name -> Node (scalar summaries, reference tf.summary.scalar)
        self.device = device or torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.seed = 42
        # host-side inputs that must be refreshed before every hipGraph replay (e.g. the k-means seeds drawn from the host RNG):
        # callables run by Network._train_graphed right before graph.replay()
        self.pre_replay_hooks = []

    @contextlib.contextmanager
    def as_default(self):
        _default_stack.append(self)
        try:
            yield self
        finally:
            _default_stack.pop()

    # ---- naming
    def scoped(self, name):
        return '/'.join(self.scope_stack + [name])

    @contextlib.contextmanager
    def variable_scope(self, name):
        self.scope_stack.append(name)
        try:
            yield
        finally:
            self.scope_stack.pop()

    # ---- variables
    def get_variable(self, name, shape=None, initializer=None):
        full = self.scoped(name)
        if full in self.variables:
            return self.variables[full]
        if initializer is None:
            raise KeyError(full)
        val = initializer(shape)
        if not torch.is_tensor(val):
            val = torch.as_tensor(val, dtype=torch.float32)
        t = val.to(device=self.device, dtype=torch.float32).contiguous()
        t.requires_grad_(False)
        t.ams_name = full
        self.variables[full] = t
        return t

    def global_variables(self):
        return list(self.variables.values())

    # ---- nodes
    def add_node(self, node):
        self.nodes[node.name] = node
        return node

    def get_tensor_by_name(self, name):
        key = name[:-2] if name.endswith(':0') else name
        return self.nodes[key]


def get_default_graph():
    if not _default_stack:
        _default_stack.append(Graph())
    return _default_stack[-1]


def reset_default_graph():
    del _default_stack[:]


class Run(object):
    """One evaluation pass (the analogue of a sess.run call): memoises node values and carries the feeds."""

    def __init__(self, feeds=None, training=True, new_pass=True):
        self.cache = {}
        self.feeds = feeds or {}
        self.training = training
        if new_pass:
            from . import ops
            ops.PASS[0] += 1            # weight bounds of the fp16x3 products are measured once per pass (ops.param_amax)
        # new_pass=False: an auxiliary Run (input probe, the front end of the NEXT batch computed inside this step): it neither counts as
        # a pass nor begins one
        self.begun = not new_pass


_RUNNING = []                           # the Runs whose nodes are being evaluated (innermost last)


def current_run():
    """The evaluation pass a node function is running in (None outside one): layers built without access to the run -- the
    reference reads `inputs/is_training:0` from the graph instead (utils/ops.py:362-363) -- ask it whether this is a training pass."""
    return _RUNNING[-1] if _RUNNING else None


class Node(object):
    def __init__(self, name, fn, graph=None, register=True):
        g = graph or get_default_graph()
        self.name = g.scoped(name) if register else name
        self.fn = fn
        if register:
            g.add_node(self)

    def value(self, run):
        k = id(self)
        if k not in run.cache:
            if not run.begun:
                run.begun = True
                if torch.cuda.is_available():
                    from . import ops, functional
                    ops.pass_begin(functional.OVERLAP.side())
            _RUNNING.append(run)
            try:
                run.cache[k] = self.fn(run)
            finally:
                _RUNNING.pop()
        return run.cache[k]

    def __repr__(self):
        return 'Node(%s)' % self.name


class Placeholder(Node):
    def __init__(self, name):
        super(Placeholder, self).__init__(name, self._get)

    def _get(self, run):
        if self not in run.feeds:
            raise KeyError('placeholder %s was not fed' % self.name)
        return run.feeds[self]


def scope(function):
    """utils/ops.py:34-44: cached lazy property that builds its node(s) once inside variable_scope(name)."""
    name = function.__name__
    attribute = '_cache_' + name

    @property
    @functools.wraps(function)
    def decorator(self):
        if not hasattr(self, attribute):
            g = get_default_graph()
            with g.variable_scope(name):
                setattr(self, attribute, function(self))
        return getattr(self, attribute)

    @decorator.setter
    def decorator(self, value):
        setattr(self, attribute, value)

    return decorator


def get_scope_variable(scope_name, var, shape=None, initializer=None):
    """utils/ops.py:61-68."""
    g = get_default_graph()
    with g.variable_scope(scope_name):
        return g.get_variable(var, shape, initializer)
