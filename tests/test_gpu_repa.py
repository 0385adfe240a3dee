"""RePaMoE compatibility: the fork's masked-SwiGLU experts (RePaMLP) and its RePaMoE wrapper, which
subclasses ``deepspeed.moe.layer.MoE``, calls ``super().forward(x, *args, **kwargs)``, adjusts the gated ratio of
every expert and, after ``reparam()``, adds a shared dense Linear to the block's output
(moellava/model/language_model/llava_stablelm_moe.py:613-773, llava_qwen_moe.py:693-856).

``_repa_moe_class(base)`` restates that wrapper over any MoE base class, so the SAME wrapper code runs over the HIP
block and over the literal oracle; the experts are ``tests.expert_mlps.RePaMLP`` (product side) and
``oracle.expert_mlps.RePaMLPRef`` (oracle side) with identical weights."""
import pytest
import torch
from torch import nn

import moe_llava_amd
from tests.expert_mlps import RePaMLP
from oracle import ds_moe_literal as lit
from oracle.expert_mlps import RePaMLPRef
from tests.test_gpu_parity import DEV

pytestmark = pytest.mark.gpu


def _repa_moe_class(base):
    class RePaMoE(base):  # llava_stablelm_moe.py:658-772
        def __init__(self, hidden_size, expert, num_experts=4, ep_size=1, k=2, capacity_factor=1.0,
                     eval_capacity_factor=1.0, min_capacity=4, use_residual=False, gated_ratio=1.0, reparamed=False,
                     **extra):
            super().__init__(hidden_size=hidden_size, expert=expert, num_experts=num_experts, ep_size=ep_size, k=k,
                             capacity_factor=capacity_factor, eval_capacity_factor=eval_capacity_factor,
                             min_capacity=min_capacity, use_residual=use_residual, **extra)
            self.gated_ratio = 1.0
            if gated_ratio < 1.0:
                self.adjust_gated_ratio(gated_ratio)
            self.reparamed = False
            self.reparam_ffn = None
            if reparamed:
                self.reparam()

        def forward(self, x, *args, **kwargs):
            if self.reparamed and self.reparam_ffn is not None:
                out, l_aux, counts = super().forward(x, *args, **kwargs)
                return out + self.reparam_ffn(x), l_aux, counts
            return super().forward(x, *args, **kwargs)

        def reparam(self):
            if self.reparamed:
                return
            ws = [e.reparam() for e in self.deepspeed_moe.experts.deepspeed_experts]
            agg = torch.stack(ws, dim=0).mean(dim=0)
            self.reparam_ffn = nn.Linear(agg.shape[0], agg.shape[1], bias=False)
            self.reparam_ffn.weight.data = agg
            self.reparam_ffn.to(ws[0].device)
            self.reparamed = True

        def adjust_gated_ratio(self, ratio):
            for e in self.deepspeed_moe.experts.deepspeed_experts:
                e.adjust_gated_ratio(ratio)
            self.gated_ratio = ratio

    return RePaMoE


def _pair(names, M, I, E, ratio, seed, cf=1.5):
    g = torch.Generator().manual_seed(seed)
    Mine, Ref = _repa_moe_class(moe_llava_amd.MoE), _repa_moe_class(lit.MoERef)
    kw = dict(num_experts=E, k=2, capacity_factor=cf, eval_capacity_factor=2.0, min_capacity=0, gated_ratio=ratio,
              top2_2nd_expert_sampling=False)
    ref = Ref(M, RePaMLPRef(M, I, names).to(torch.bfloat16), **kw)
    for e in ref.deepspeed_moe.experts.deepspeed_experts:
        for p in e.parameters():
            p.data = (torch.randn(p.shape, generator=g) * 0.05).to(torch.bfloat16)
    ref.deepspeed_moe.gate.wg.weight.data = torch.randn(E, M, generator=g) * 0.1
    mine = Mine(M, RePaMLP(M, I, names).to(torch.bfloat16), **kw)
    res = mine.load_state_dict(ref.state_dict(), strict=True)       # same keys incl. the 'mask' buffers
    assert not res.missing_keys and not res.unexpected_keys
    mine = mine.to(DEV)
    return ref.train(), mine.train(), g


def _cmp(name, got, want, tol):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    print(f"[repa parity] {name}: max_abs={err:.3e} ref_max={scale:.3e} rel={err / max(scale, 1e-30):.3e}")
    assert err <= tol * scale + 1e-6, name


@pytest.mark.parametrize("names,M,I,E,ratio", [
    ("stablelm", 128, 320, 4, 0.5),      # n_gated = 160: inside a 16-column fragment group
    ("stablelm", 64, 128, 4, 0.25),
    ("qwen", 128, 192, 4, 0.7),          # n_gated = 134: not a multiple of 4
    ("stablelm", 128, 256, 2, 0.0),      # nothing gated: h = up(x)
])
def test_masked_swiglu_forward_and_backward(names, M, I, E, ratio):
    ref, mine, g = _pair(names, M, I, E, ratio, seed=21)
    assert mine.deepspeed_moe.experts.deepspeed_experts[0].num_gated_channels == int(I * ratio)
    x = torch.randn(2, 150, M, generator=g).to(torch.bfloat16)
    R = torch.randn(2, 150, M, generator=g).to(torch.bfloat16)
    # inference path (fused epilogue)
    with torch.no_grad():
        o_inf, _, _ = mine(x.to(DEV))
    # training path + gradients
    xg = x.to(DEV).requires_grad_(True)
    out, l_aux, cnt = mine(xg)
    assert mine.deepspeed_moe.experts.n_gated == int(I * ratio)
    assert torch.equal(out.detach(), o_inf)
    ((out.float() * R.to(DEV).float()).sum() + 0.3 * l_aux).backward()
    torch.cuda.synchronize()
    ref.deepspeed_moe.gate.logits_override = mine.deepspeed_moe.gate.last_logits.cpu()
    xr = x.clone().requires_grad_(True)
    r_out, r_laux, r_cnt = ref(xr)
    ((r_out.float() * R.float()).sum() + 0.3 * r_laux).backward()
    tag = f"{names} M{M} I{I} ratio={ratio}"
    assert torch.equal(cnt.cpu(), r_cnt)
    _cmp(tag + " out", out, r_out, 1e-2)
    _cmp(tag + " dx", xg.grad, xr.grad, 2e-2)
    _cmp(tag + " dwg", mine.deepspeed_moe.gate.wg.weight.grad, ref.deepspeed_moe.gate.wg.weight.grad, 2e-2)
    rp = dict(ref.deepspeed_moe.experts.named_parameters())
    for n, p in mine.deepspeed_moe.experts.named_parameters():
        _cmp(f"{tag} {n}", p.grad, rp[n].grad, 2e-2)
    # the SiLU'd projection gets no gradient on masked channels
    n_g = int(I * ratio)
    gname = "gate_proj" if names == "stablelm" else "w2"
    gp = getattr(mine.deepspeed_moe.experts.deepspeed_experts[0], gname).weight.grad
    assert float(gp[n_g:].float().abs().max()) == 0.0


@pytest.mark.parametrize("names,M,I,ratio", [("stablelm", 128, 320, 0.55), ("qwen", 64, 256, 0.5)])
def test_reparam_shrinks_experts_and_adds_dense_branch(names, M, I, ratio):
    """After reparam(): experts are ordinary SwiGLU of width int(I*ratio) (176: padded to 192 inside the packed
    layout), parameters are new tensors (re-pack), and the wrapper adds reparam_ffn(x)."""
    ref, mine, g = _pair(names, M, I, 4, ratio, seed=33)
    x = torch.randn(1, 200, M, generator=g).to(torch.bfloat16)
    with torch.no_grad():
        before, _, _ = mine(x.to(DEV))
    mine.reparam()
    ref.reparam()
    n = int(I * ratio)
    e0 = mine.deepspeed_moe.experts.deepspeed_experts[0]
    assert e0.reparamed and mine.reparam_ffn.weight.shape == (M, M)
    xg = x.to(DEV).requires_grad_(True)
    out, l_aux, cnt = mine(xg)
    assert mine.deepspeed_moe.experts.n_gated == -1
    w1 = mine.deepspeed_moe.experts._packed["w1"]
    assert w1.shape[1] == 2 * ((n + 63) // 64 * 64)
    out.float().sum().backward()
    torch.cuda.synchronize()
    ref.deepspeed_moe.gate.logits_override = mine.deepspeed_moe.gate.last_logits.cpu()
    xr = x.clone().requires_grad_(True)
    r_out, _, r_cnt = ref(xr)
    r_out.float().sum().backward()
    assert torch.equal(cnt.cpu(), r_cnt)
    _cmp(f"{names} reparamed out", out, r_out, 1e-2)
    _cmp(f"{names} reparamed dx", xg.grad, xr.grad, 2e-2)
    rp = dict(ref.named_parameters())
    for nme, p in mine.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, nme
        _cmp(f"{names} reparamed {nme}", p.grad, rp[nme].grad, 2e-2)
    # padding of the packed layout stays zero
    if n % 64:
        Ip = (n + 63) // 64 * 64
        assert float(w1[:, n:Ip].float().abs().max()) == 0.0
    assert out.shape == before.shape
