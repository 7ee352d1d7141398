"""Host side of the soft k-means backward: picks the SELECTED try's centroid trace and hands it to ONE C call that
replays the unrolled iterations in reverse (csrc/kmeans_soft.hip)."""
import torch

from . import ops
from ._lib import load, check


def soft_bwd(xn, inv, init_idx, best, w, trace, dsel, dout, C, tries, iterations, beta, assign_at_end, faithful_tile, inv0=None):
    """Returns d loss / d X (pre-normalisation input [b,L,E]; with inv0: the input of the normalisation BEFORE that one, see
    functional.KMeansSoft).  trace = [c_0, c_1, den_0, c_2, den_1, ...] over all R rows."""
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
    # ONE call: final-assignment pass, the iteration passes in reverse, the dx pass (csrc/kmeans_soft.hip)
    nb = lib.ams_kmeans_soft_bwd_workspace_bytes(b, L, E, C, iterations)
    ws = ops._ws(nb, xn)
    p, s = ops._p, ops._s
    w_final = None if assign_at_end else wsel
    # ops.kmeans_run keeps the trace as slices of two stacked buffers: with one try per utterance they ARE what the call wants
    def stacked(ts):
        base = ts[0]._base
        if (index is None and base is not None and base.is_contiguous() and base.shape[0] == len(ts) and base.shape[1:] == ts[0].shape
                and all(t._base is base and t.data_ptr() == base[i].data_ptr() for i, t in enumerate(ts))):
            return base
        return torch.stack(ts).contiguous()
    cst = stacked(cents)                                                    # [n_it + 1, b, C, E]
    dst = stacked(dens) if iterations else None                             # [n_it, b, C]
    dxn = torch.empty_like(xn)
    g = torch.empty((b, C, E), dtype=torch.float32, device=dev)
    dsel_c = dsel.contiguous() if dsel is not None else None
    dout_c = dout.contiguous() if dout is not None else None
    # c_0 = xn[idx]: the remaining centroid gradient goes onto the picked points, and -- when the input was normalised here -- every row
    # through the l2-normalise Jacobian, both inside the same call (the pass that writes dx holds the point in registers)
    idx_sel = (init_idx if index is None else init_idx[index]).to(torch.int32).contiguous()      # [b, C]
    check(lib.ams_kmeans_soft_bwd(p(xn), p(wsel), p(w_final), p(cst), p(dst), p(dsel_c), p(dout_c), p(inv), p(inv0), p(idx_sel), p(dxn), p(g),
                                  b, L, E, C, float(beta), iterations, p(ws), nb, s()), 'ams_kmeans_soft_bwd')
    # the pass that wrote dx left max |dx| in the workspace (ABI 5): the bound of the dense layer's two gradient products
    off = lib.ams_kmeans_soft_bwd_amax_offset(b, L, E, C, iterations) // 4
    return ops.tag_amax(dxn, ws[off:off + 1])
