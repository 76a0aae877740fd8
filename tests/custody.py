"""Chain of custody between the HIP path and the oracle (shared by the parity tests): "indices bit-exact" and "values
within 1e-4" cannot both hold end to end -- a neighbour choice on an fp32 near-tie flips when the features it is computed
from differ in the last bits -- so the claim is split where it is exact: the head's index decisions on the HIP node
matrix equal the oracle's on the same matrix, and with them injected every logit agrees within 1e-4 at EVERY point."""
import torch

from oracle import r3d_oracle as O

TOL = 1e-4


def close(a, b):
    """max |a - b| / max(1, |b|)"""
    return ((a - b).abs() / b.abs().clamp(min=1.0)).max().item()


def head_custody(m, cfg, sd, data, logits, loss, eval_flag=False):
    """m: model after a forward with m._trace set; data: the CPU episode.  Asserts the custody of the head."""
    from r3dfsseg_amd import ops
    sx, sy, qx, qy = data[:4]
    n_way, k_shot, N = cfg["n_way"], cfg["k_shot"], cfg["pc_npts"]
    Sn = n_way * k_shot
    tr, hb = m._trace, m._head[1]
    sfeat_cm = tr["sfeat"].cpu().reshape(Sn, N, -1).transpose(1, 2).contiguous()
    qfeat_cm = tr["qfeat"].cpu().reshape(qx.shape[0], N, -1).transpose(1, 2).contiguous()
    sf4 = sfeat_cm.view(n_way, k_shot, -1, N)
    pl = None
    if eval_flag:
        pl, clean_flag = O.mean_pl_support_y_multi_scale(sf4, sy, sx.reshape(n_way, k_shot, cfg["pc_in_dim"], N))
        assert torch.equal(tr["shot_keep"].cpu().view(n_way, k_shot).float(), clean_flag)
    fg_p, fg_l, _, _ = O.get_foreground_prototypes(sf4, sy, cfg["n_subprototypes"], n_way + 1, pl)
    bg_p, bg_l, _, _ = O.get_background_prototypes(sf4, torch.logical_not(sy), cfg["n_subprototypes"], n_way + 1)
    protos = torch.cat((bg_p, fg_p), 0)
    n_proto = int(hb.desc.view(-1, 32)[0, ops.HD_N_PROTO].item())
    n = int(hb.desc.view(-1, 32)[0, ops.HD_N_NODES].item())
    assert n_proto == protos.shape[0] and n == n_proto + qx.shape[0] * N
    nodes = hb.nodes[:n].cpu()
    assert close(nodes[:n_proto], protos) <= 1e-5   # same FPS seeds, same assignment, same cluster means
    assert torch.equal(nodes[n_proto:], qfeat_cm.transpose(1, 2).reshape(-1, sfeat_cm.shape[1]))
    nbr_hip = tr["nbr"].reshape(-1, hb.kp1)[:n].cpu().to(torch.int64)
    assert torch.equal(nbr_hip, O.knn_l2(nodes, hb.kp1))                      # 201-NN lists: bit-exact on the HIP nodes
    A = O.affinity(nodes, cfg["k_connect"], cfg["sigma"])
    # label columns: float4 per node, for more than 3 ways two planes of 4 (classes 4..7 behind the E systems of plane 0)
    pl = hb.E * hb.n_cap
    cols = lambda t: torch.cat([t[p * pl:p * pl + n] for p in range(hb.planes)], 1)[:, :n_way + 1].cpu()
    Yh = cols(hb.Y)
    assert torch.equal(Yh[:n_proto], torch.cat((bg_l, fg_l), 0)) and not Yh[n_proto:].any()   # one-hot rows as mpti.py:505-507
    Zo = O.label_propagate(A, Yh)
    assert close(cols(hb.Z), Zo) <= TOL
    want = Zo[n_proto:].view(-1, N, n_way + 1).transpose(1, 2)
    assert close(logits.cpu(), want) <= TOL                                    # every point, no fraction of outliers
    assert abs(loss.item() - torch.nn.functional.cross_entropy(want, qy).item()) <= TOL
