"""Host orchestration of the soft k-means backward: replays the unrolled iterations of the
SELECTED try in reverse, one streaming HIP pass each (csrc/kmeans.hip: kmeans_soft_bwd_kernel)."""
import torch

from . import ops
from ._lib import load, check


def soft_bwd(xn, inv, init_idx, best, w, trace, dsel, dout, C, tries, iterations, beta, assign_at_end, faithful_tile):
    """Returns d loss / d X (pre-normalisation input [b,L,E]).  trace = [c_0, c_1, den_0, c_2, den_1, ...] over all R rows."""
    lib = load()
    b, L, E = xn.shape
    dev = xn.device
    if tries == 1:
        # one try per utterance: row r IS utterance r -- no gathers (they were 2 launches per unrolled iteration)
        index = None
        pick = lambda t: t                                                  # noqa: E731
    else:
        index = best.long() + torch.arange(b, device=dev) * tries
        pick = lambda t: t[index].contiguous()                              # noqa: E731
    cents = [pick(trace[0])]
    dens = []
    for i in range(iterations):
        cents.append(pick(trace[1 + 2 * i]))
        dens.append(pick(trace[2 + 2 * i]))
    wsel = None
    if w is not None:
        if index is None:
            wsel = w
        else:
            wrow = (index % b) if faithful_tile else (index // tries)
            wsel = w[wrow].contiguous()
    nb = lib.ams_kmeans_workspace_bytes(b, L, E, C)
    ws = ops._ws(nb, xn)
    # G[i] = d loss / d c_i: the final pass accumulates into G[iterations], iteration i reads G[i+1] and writes G[i] -- the
    # per-iteration gradients phase 2 needs are then the slice G[1:], with no copies in between
    G = torch.empty((iterations + 1, b, C, E), dtype=torch.float32, device=dev)
    if dsel is not None:
        G[iterations].copy_(dsel)
    else:
        G[iterations].zero_()
    p, s = ops._p, ops._s
    w_final = None if assign_at_end else wsel
    # phase 1: centroid gradients only (dx == NULL): one read of xn per pass, no read-modify-write of dx
    if dout is not None:
        dout = dout.contiguous()
        check(lib.ams_kmeans_soft_bwd_pass(p(xn), p(w_final), p(cents[-1]), p(None), p(None), p(None), p(dout), p(None), p(G[iterations]),
                                           b, L, E, C, float(beta), 0, p(ws), nb, s()), 'ams_kmeans_soft_bwd_pass(final)')
    for i in range(iterations - 1, -1, -1):
        check(lib.ams_kmeans_soft_bwd_pass(p(xn), p(wsel), p(cents[i]), p(cents[i + 1]), p(dens[i]), p(G[i + 1]), p(None), p(None), p(G[i]),
                                           b, L, E, C, float(beta), 1, p(ws), nb, s()), 'ams_kmeans_soft_bwd_pass(iter)')
    g = G[0]
    gs = G[1:]
    # phase 2: dx of the final assignment and of every iteration in ONE pass over xn
    dxn = torch.empty_like(xn)
    cst = torch.stack(cents).contiguous()
    dst = torch.stack(dens).contiguous() if iterations else None
    check(lib.ams_kmeans_soft_bwd_dx(p(xn), p(wsel), p(w_final), p(cst), p(gs if iterations else None), p(dst), p(dout), p(dxn),
                                     b, L, E, C, float(beta), iterations, s()), 'ams_kmeans_soft_bwd_dx')
    # c_0 = xn[idx]: scatter-add the remaining centroid gradient onto the picked points (tiny: b*C rows)
    # (the C picks of a row are distinct -- np.random.choice without replacement, Kmeans_2.py:63 -- so no two updates collide)
    idx_sel = (init_idx if index is None else init_idx[index]).long()   # [b, C]
    flat = (idx_sel + torch.arange(b, device=dev).unsqueeze(1) * L).reshape(-1)
    dxn.view(b * L, E).index_add_(0, flat, g.reshape(b * C, E))
    if inv is None:
        return dxn
    return ops.l2norm_bwd(xn.view(b, L * E), inv, dxn.view(b, L * E), E).view(b, L, E)
