"""IR-Net attention kernels (csrc/relation.hip: mmt_relation_attention_{fwd,bwd}, mmt_ciam_{fwd,bwd}) against the CPU oracle
(oracle/irnet.py: `relation_module`, `ciam` -- restatements of the reference's relation_module.py:33-90 and
mask_relation_module.py:199-242, pinned to the reference by tests/golden/model160_irnet.npz), forward and every gradient, and
against the product's own tensor formulation (the library-GEMM path that the kernels replace, kept for CPU tensors).
Tolerances: fp32 sums in another order -- 2e-5 of the tensor's largest magnitude forward, 2e-4 for gradients."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what, floor=1e-12):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(b.abs().max().item(), floor)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, "%s: max error %.3e of the largest magnitude %.3e (tolerance %.1e)" % (what, err, scale, tol)


def _relation(N, C, topk, seed):
    from maskrcnn_benchmark.modeling.relation.relation_module import RelationModule
    torch.manual_seed(seed)
    mod = RelationModule(128, geo_feature_dim=64, fc_dim=(64, 16), group=16, dim=(1024, 1024, 128), topk=topk).cuda()
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * (0.08 if p.dim() > 1 else 0.05))
    f_a = torch.randn(N, C, 128, device="cuda")
    ang = torch.rand(C, N, N, 32, device="cuda") * 6.0
    pos = torch.cat((torch.sin(ang), torch.cos(ang)), -1)      # what extract_multi_position_matrix produces: sines and cosines
    return mod, f_a, pos


@pytest.mark.parametrize("N,C,topk", [(90, 4, 40), (20, 2, 40), (100, 3, 40), (64, 1, 10), (1, 2, 40), (128, 1, 40), (9, 2, 4)])
def test_relation_attention_vs_oracle(N, C, topk):
    from oracle import irnet as oi
    mod, f_a, pos = _relation(N, C, topk, seed=N)
    sd = {"r." + k: v.detach().double().cpu().requires_grad_(True) for k, v in mod.state_dict().items()}
    cfg = types.SimpleNamespace(group=16, hid=(1024, 1024, 128), geo_dim=64, rel_topk=topk)
    fa_o = f_a.detach().double().cpu().requires_grad_(True)
    ref = oi.relation_module(sd, "r.", fa_o, pos.detach().double().cpu(), cfg)
    gout = torch.randn(N, C, 128, dtype=torch.float64)
    ref.backward(gout)

    fa_g = f_a.clone().requires_grad_(True)
    out = mod(fa_g, pos)
    assert out.shape == (N, C, 128)
    _close(out, ref, 2e-5, "relation attention forward")
    out.backward(gout.float().cuda())
    _close(fa_g.grad, fa_o.grad, 2e-4, "d f_a")
    for k, p in mod.named_parameters():
        # (d WK.bias is zero in exact arithmetic -- a key bias shifts every score of a row alike and the softmax ignores it --
        # so its reference magnitude is rounding noise: gradients are measured against at least 1e-2, they are O(1..100) here)
        _close(p.grad, sd["r." + k].grad.view_as(p.grad), 2e-4, "d " + k, floor=1e-2)

    # the formulation with batched library GEMMs / top-k / scatter on the same parameters (tests/tensor_formulations.py) agrees too
    import tensor_formulations as tf
    for p in mod.parameters():
        p.grad = None
    fa_t = f_a.clone().requires_grad_(True)
    out_t = tf.relation_attention(mod, fa_t, pos)
    _close(out, out_t, 2e-5, "kernel vs tensor formulation")
    out_t.backward(gout.float().cuda())
    _close(fa_g.grad, fa_t.grad, 2e-4, "d f_a, kernel vs tensor formulation")


def test_relation_attention_tie_goes_to_the_lower_index():
    """two identical boxes: identical score columns -- the top-k cut keeps the lower index, like a stable descending sort"""
    from maskrcnn_benchmark import _hip as H
    C, N, G, DQ, DV, topk = 1, 8, 2, 4, 2, 3
    torch.manual_seed(0)
    q = torch.randn(C * N, G * DQ, device="cuda")
    k = torch.randn(C * N, G * DQ, device="cuda")
    k[5] = k[2]
    wg = torch.ones(C * N * N, G, device="cuda")
    v = torch.randn(C * N, G * DV, device="cuda")
    out, P = H.relation_attention_fwd(q, k, wg, v, torch.zeros(G * DV, device="cuda"), C, N, G, topk, 0.5)
    S = 0.5 * torch.einsum("ngd,mgd->gnm", q.view(N, G, DQ), k.view(N, G, DQ))
    order = torch.sort(S, dim=2, descending=True, stable=True)[1][:, :, :topk]
    want = torch.zeros_like(S).scatter(2, order, torch.softmax(torch.gather(S, 2, order), 2))
    assert ((P[0] > 0) == (want > 0)).all()
    _close(P[0], want, 1e-5, "top-k softmax")
    assert ((P[0] > 0).sum(2) == topk).all()


@pytest.mark.parametrize("sizes", [[5, 1, 37, 70], [128], [1, 1, 1], [200, 56]])
def test_ciam_vs_oracle(sizes):
    from oracle import irnet as oi
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.relation.mask_relation_module import CIAM_Module
    torch.manual_seed(len(sizes) * 7 + sizes[0])
    n, C = sum(sizes), 16
    mod = CIAM_Module(make_default_cfg()).cuda()
    with torch.no_grad():
        mod.gamma.fill_(0.7)
    x = torch.relu(torch.randn(n, C, 14, 14, device="cuda") * 0.3 + 0.1).contiguous(memory_format=torch.channels_last)
    group = torch.cat([torch.full((s,), 3 * i + 1, dtype=torch.int64) for i, s in enumerate(sizes)]).cuda()
    gout = torch.randn(n, C, 14, 14, dtype=torch.float64)

    # oracle: the reference calls the module once per (image, class) slice
    xo = x.detach().double().cpu().contiguous().requires_grad_(True)
    go = torch.tensor(0.7, dtype=torch.float64, requires_grad=True)
    parts, st = [], 0
    for s in sizes:
        parts.append(oi.ciam(go, xo[st:st + s], None))
        st += s
    ref = torch.cat(parts)
    ref.backward(gout)

    xg = x.clone().requires_grad_(True)
    out = mod(xg, group)
    _close(out, ref, 2e-5, "CIAM forward")
    out.backward(gout.float().cuda())
    _close(xg.grad, xo.grad, 2e-4, "d x")
    _close(mod.gamma.grad, go.grad.view(1), 2e-4, "d gamma")

    import tensor_formulations as tf
    mod.gamma.grad = None
    xt = x.clone().requires_grad_(True)
    out_t = tf.ciam(mod.gamma, xt, group)
    _close(out, out_t, 2e-5, "kernel vs tensor formulation")
    out_t.backward(gout.float().cuda())
    _close(xg.grad, xt.grad, 2e-4, "d x, kernel vs tensor formulation")


def test_ciam_dead_channel():
    """an instance whose channel is all zero after the ReLU: every energy of that row ties at 0 and the arg-max is a choice
    (the kernel takes the first column); the gradient that choice moves lands on the zero channel itself, where the ReLU in
    front of CIAM masks it -- compared under that mask"""
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.relation.mask_relation_module import CIAM_Module
    torch.manual_seed(3)
    n, C = 12, 16
    mod = CIAM_Module(make_default_cfg()).cuda()
    with torch.no_grad():
        mod.gamma.fill_(-0.4)
    x = torch.relu(torch.randn(n, C, 14, 14, device="cuda") * 0.3 + 0.1)
    x[3, 2] = 0
    x[7, 11] = 0
    group = torch.zeros(n, dtype=torch.int64, device="cuda")
    gout = torch.randn(n, C, 14, 14, device="cuda")
    xg = x.clone().requires_grad_(True)
    mod(xg, group).backward(gout)
    import tensor_formulations as tf
    xt = x.clone().requires_grad_(True)
    tf.ciam(mod.gamma, xt, group).backward(gout)
    m = (x > 0).float()
    _close(xg.grad * m, xt.grad * m, 2e-4, "d x under the ReLU mask")
