"""Oracle, timed form: the front_DPCL training step as a torch-CPU program (oneDNN / MKL kernels, all host threads).

Test infrastructure only -- see oracle/__init__.py.  This is the `cpu_baseline` of bench.py ("the build's own fp32 CPU
restatement ... torch-CPU/oneDNN formulation of the 8(a) rows, which is what TF-CPU/Eigen+MKL would also reduce to").
It is NOT TensorFlow and not a target: the reference's own CPU path cannot run here (Python 2 + TF 1.4).  It follows the same
reference lines as oracle/step.py::front_dpcl_loss and is held to it in tests/test_oracle_core.py:

  front      Adapt.front path A, frozen          models/adapt.py:95-126        F.conv1d on the SAME-padded waveforms
  masks      one_hot(argmax_s |X_non_mix|)       models/network.py:369-378
  BLSTM x3   dynamic_rnn(BasicLSTMCell), 2 dirs  utils/ops.py:358-383          torch.nn.LSTM (fused CPU kernel) with the TF kernel
                                                                               re-laid out: gates i,j,f,o -> i,f,g,o, forget bias +1
  dense      conv1d k=1 + l2_normalize over E    utils/ops.py:486-503,323-324
  cost       DPCL, un-squared Frobenius norms    models/dpcl.py:41-87
  update     'Adam' = AMSGrad(b2=.99, eps=1e-3)  models/network.py:181-182, utils/ops.py:686-703
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import front as ofront
from .step import lstm_names


def _tf_to_torch_lstm(K, b, D):
    """TF BasicLSTMCell kernel [D+H, 4H] (gate order i, j, f, o; forget_bias 1.0 added at run time) -> torch.nn.LSTM's
    weight_ih [4H, D], weight_hh [4H, H], bias (gate order i, f, g, o; the +1 folded into b_f)."""
    H = K.shape[1] // 4
    i, j, f, o = (K[:, k * H:(k + 1) * H] for k in range(4))
    Kt = torch.cat([i, f, j, o], 1)
    bi, bj, bf_, bo = (b[k * H:(k + 1) * H] for k in range(4))
    bt = torch.cat([bi, bf_ + 1.0, bj, bo])
    return Kt[:D].t().contiguous(), Kt[D:].t().contiguous(), bt


class FrontDPCLStep(object):
    """Holds the parameters as torch tensors (TF variable names) and runs whole steps on the CPU."""

    def __init__(self, P, hop, nb_layers, E, lr=1e-3, dtype=torch.float32):
        self.hop, self.NL, self.E, self.lr = hop, nb_layers, E, lr
        self.P = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in P.items()}
        self.names = sorted(n for n in self.P if n.startswith('prediction/'))
        for n in self.names:
            self.P[n].requires_grad_(True)
        self.m = {n: torch.zeros_like(self.P[n]) for n in self.names}
        self.v = {n: torch.zeros_like(self.P[n]) for n in self.names}
        self.vhat = {n: torch.zeros_like(self.P[n]) for n in self.names}
        self.b1p, self.b2p = 0.9, 0.99

    def loss(self, x_mix, x_non_mix):
        P, hop, E = self.P, self.hop, self.E
        B, S, L = x_non_mix.shape
        with torch.no_grad():                                       # frozen front (utils/trainer.py:587-588)
            x = torch.cat([x_mix, x_non_mix.reshape(B * S, L)], 0)
            f = P['front/window/w'].abs()[:, None] * P['front/bases/bases']        # adapt.py:106
            W = f.shape[0]
            T, pl, pr = ofront.same_pads(L, W, hop)
            y = F.conv1d(F.pad(x[:, None, :], (pl, pr)), f.t()[:, None, :], stride=hop).transpose(1, 2)     # [Bt, T, N]
            X = y[:B]
            Xnm = y[B:].reshape(B, S, T, -1).abs()
            Y = F.one_hot(Xnm.argmax(1), S).to(y.dtype)                             # [B, T, F, S]
        h = X
        for i in range(self.NL):
            kf, bf, kb, bb = (P[n] for n in lstm_names('prediction', i))
            D = h.shape[-1]
            H = kf.shape[1] // 4
            wf = _tf_to_torch_lstm(kf, bf, D)
            wb = _tf_to_torch_lstm(kb, bb, D)
            zeros = torch.zeros(4 * H, dtype=h.dtype)
            flat = [wf[0], wf[1], wf[2], zeros, wb[0], wb[1], wb[2], zeros]
            hx = (torch.zeros(2, B, H, dtype=h.dtype), torch.zeros(2, B, H, dtype=h.dtype))
            h = torch._VF.lstm(h, hx, flat, True, 1, 0.0, False, True, True)[0]    # has_biases, 1 layer, bidirectional, batch_first
        u = h @ P['prediction/W'] + P['prediction/b']                               # [B, T, F*E]
        u = u.reshape(B, -1, E)
        V = u * torch.rsqrt(torch.clamp((u * u).sum(-1, keepdim=True), min=1e-12))
        Yf = Y.reshape(B, -1, S)
        # dpcl.py:54-80: D = 1/sqrt(Y (Y^T 1)); cost = mean_b(|V^T D V|_F - 2 |V^T D Y|_F + |Y^T D Y|_F)
        cnt = Yf.sum(1)                                                              # [B, S]
        d = torch.rsqrt((Yf * cnt[:, None, :]).sum(-1))                              # [B, TF]
        DV, DY = V * d[:, :, None], Yf * d[:, :, None]
        t1 = torch.linalg.matrix_norm(V.transpose(1, 2) @ DV)
        t2 = torch.linalg.matrix_norm(V.transpose(1, 2) @ DY)
        t3 = torch.linalg.matrix_norm(Yf.transpose(1, 2) @ DY)
        return (t1 - 2.0 * t2 + t3).mean()

    def step(self, x_mix, x_non_mix):
        """forward + backward + AMSGrad update; returns the cost."""
        for n in self.names:
            self.P[n].grad = None
        cost = self.loss(x_mix, x_non_mix)
        cost.backward()
        lr_t = self.lr * np.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        with torch.no_grad():
            for n in self.names:
                g = self.P[n].grad
                self.m[n].mul_(0.9).add_(g, alpha=0.1)
                self.v[n].mul_(0.99).addcmul_(g, g, value=0.01)
                torch.maximum(self.vhat[n], self.v[n], out=self.vhat[n])
                self.P[n].addcdiv_(self.m[n], self.vhat[n].sqrt().add_(1e-3), value=-lr_t)
        self.b1p *= 0.9
        self.b2p *= 0.99
        return float(cost.detach())
