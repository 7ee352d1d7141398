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
    index = best.long() + torch.arange(b, device=dev) * tries
    cents = [trace[0][index].contiguous()]
    dens = []
    for i in range(iterations):
        cents.append(trace[1 + 2 * i][index].contiguous())
        dens.append(trace[2 + 2 * i][index].contiguous())
    wsel = None
    if w is not None:
        wrow = (index % b) if faithful_tile else (index // tries)
        wsel = w[wrow].contiguous()
    nb = lib.ams_kmeans_workspace_bytes(b, L, E, C)
    ws = ops._ws(nb, xn)
    g = dsel.contiguous().clone() if dsel is not None else torch.zeros((b, C, E), dtype=torch.float32, device=dev)
    p, s = ops._p, ops._s
    w_final = None if assign_at_end else wsel
    # phase 1: centroid gradients only (dx == NULL): one read of xn per pass, no read-modify-write of dx
    if dout is not None:
        dout = dout.contiguous()
        check(lib.ams_kmeans_soft_bwd_pass(p(xn), p(w_final), p(cents[-1]), p(None), p(None), p(None), p(dout), p(None), p(g),
                                           b, L, E, C, float(beta), 0, p(ws), nb, s()), 'ams_kmeans_soft_bwd_pass(final)')
    gs = torch.empty((max(iterations, 1), b, C, E), dtype=torch.float32, device=dev)
    for i in range(iterations - 1, -1, -1):
        gs[i].copy_(g)
        g_new = torch.empty_like(g)
        check(lib.ams_kmeans_soft_bwd_pass(p(xn), p(wsel), p(cents[i]), p(cents[i + 1]), p(dens[i]), p(gs[i]), p(None), p(None), p(g_new),
                                           b, L, E, C, float(beta), 1, p(ws), nb, s()), 'ams_kmeans_soft_bwd_pass(iter)')
        g = g_new
    # phase 2: dx of the final assignment and of every iteration in ONE pass over xn
    dxn = torch.empty_like(xn)
    cst = torch.stack(cents).contiguous()
    dst = torch.stack(dens).contiguous() if iterations else None
    check(lib.ams_kmeans_soft_bwd_dx(p(xn), p(wsel), p(w_final), p(cst), p(gs if iterations else None), p(dst), p(dout), p(dxn),
                                     b, L, E, C, float(beta), iterations, s()), 'ams_kmeans_soft_bwd_dx')
    # c_0 = xn[idx]: scatter-add the remaining centroid gradient onto the picked points (tiny: b*C rows)
    idx_sel = init_idx[index].long()                                   # [b, C]
    rows = torch.arange(b, device=dev).unsqueeze(1).expand(b, C)
    dxn.index_put_((rows.reshape(-1), idx_sel.reshape(-1)), g.reshape(b * C, E), accumulate=True)
    if inv is None:
        return dxn
    return ops.l2norm_bwd(xn.view(b, L * E), inv, dxn.view(b, L * E), E).view(b, L, E)
