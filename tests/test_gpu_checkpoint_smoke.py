"""End-to-end smoke of the drop-in boundary, config-only / random-init because no checkpoint
is available offline: a small decoder stack is converted to MoE the way the reference does it
(``initialize_moe_modules``, llava_stablelm_moe.py:548-564: ``layer.mlp = MoE(hidden, expert=layer.mlp, ...)`` on the
layers of ``moe_layers_idx`` + the weight-copy assert), saved, re-created the way ``EvalMoELLaVAStablelmForCausalLM``
does (:589-600: MoE modules built first, then ``load_state_dict``), hooked by name like
``get_gating_logit_by_hook`` (moellava/utils.py:32-46) and run through the decoder-layer glue of :163-170
(3-tuple unpack, residual add, ``moe_loss = coef * sum(l_aux)``, :461-472) -- with ``MoE`` imported from the
``deepspeed.moe.layer`` path through the shim.  The same stack with the literal oracle's MoERef on the CPU, fed the
captured gate logits layer by layer, is the checker.  Not the reference's model code (attention is replaced by a
token-wise mixer: everything outside the MoE block is out of scope)."""
import copy
import io
import sys

import pytest
import torch
from torch import nn

from oracle import ds_moe_literal as lit
from oracle.expert_mlps import StableLMMLPRef
from tests.test_gpu_parity import DEV

pytestmark = pytest.mark.gpu

M, I, E, LAYERS, MOE_IDX = 128, 256, 4, 4, (1, 3)


class _Layer(nn.Module):
    def __init__(self, mlp):
        super().__init__()
        self.input_layernorm = nn.LayerNorm(M)
        self.mixer = nn.Linear(M, M, bias=False)          # stands in for self_attn (out of scope)
        self.post_attention_layernorm = nn.LayerNorm(M)
        self.mlp = mlp

    def forward(self, h):
        h = h + self.mixer(self.input_layernorm(h))
        residual = h
        out = self.mlp(self.post_attention_layernorm(h))
        moe_losses = []
        if len(out) == 3:                                   # llava_stablelm_moe.py:167-169
            moe_losses.append(out[1])
            out = out[0]
        return residual + out, moe_losses


class _Stack(nn.Module):
    def __init__(self, mlp_cls):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(mlp_cls(M, I)) for _ in range(LAYERS)])

    def to_moe(self, moe_cls, **kw):                        # initialize_moe_modules, :548-564
        for idx in MOE_IDX:
            pretrained = self.layers[idx].mlp.state_dict()
            self.layers[idx].mlp = moe_cls(M, expert=self.layers[idx].mlp, num_experts=E, ep_size=1, k=2,
                                           capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0,
                                           use_residual=False, **kw)
            for e in self.layers[idx].mlp.deepspeed_moe.experts.deepspeed_experts:
                loaded = e.state_dict()
                assert all(torch.allclose(pretrained[k], v) for k, v in loaded.items())
                assert all(torch.allclose(loaded[k], v) for k, v in pretrained.items())
        return self

    def forward(self, h):
        losses = []
        for layer in self.layers:
            h, l = layer(h)
            losses.extend(l)
        return h, losses


def _hooks(model):                                          # moellava/utils.py:32-46
    feas = []
    for n, m in model.named_modules():
        if "wg" in n and isinstance(m, nn.Linear):
            box = {}
            m.register_forward_hook(lambda mod, fin, fout, box=box: box.__setitem__("fea", fout.detach().cpu()))
            feas.append(box)
    return feas


def test_convert_save_reload_hook_and_run():
    from moe_llava_amd import shim
    from tests.expert_mlps import StableLMMLP
    saved_mods = {k: v for k, v in sys.modules.items() if k == "deepspeed" or k.startswith("deepspeed.")}
    try:
        shim.install()
        from deepspeed.moe.layer import MoE                 # the reference's import line (:29)
        torch.manual_seed(3)
        # --- "training side": dense stack -> MoE conversion -> checkpoint bytes
        trained = _Stack(StableLMMLP).to(torch.bfloat16).to_moe(MoE, top2_2nd_expert_sampling=False)
        for n, p in trained.named_parameters():
            if "deepspeed_experts" in n:                     # make the experts differ, as after MoE-tuning
                p.data += 0.02 * torch.randn_like(p)
        keys = set(trained.state_dict().keys())
        assert f"layers.1.mlp.deepspeed_moe.gate.wg.weight" in keys
        assert f"layers.3.mlp.deepspeed_moe.experts.deepspeed_experts.{E - 1}.down_proj.weight" in keys
        assert "layers.0.mlp.gate_proj.weight" in keys
        n_expert_params = sum(p.numel() for n, p in trained.named_parameters() if "deepspeed_experts" in n)
        assert n_expert_params == len(MOE_IDX) * E * 3 * M * I   # eval/model_vqa_loader.py:240 counts by this substring
        buf = io.BytesIO()
        torch.save(trained.state_dict(), buf)
        # --- "eval side": build MoE modules first, then load (EvalMoE*, :589-600)
        buf.seek(0)
        sd = torch.load(buf, weights_only=True)
        model = _Stack(StableLMMLP).to(torch.bfloat16).to_moe(MoE, top2_2nd_expert_sampling=False)
        res = model.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        model = model.to(DEV).eval()
        feas = _hooks(model)
        assert len(feas) == len(MOE_IDX)
        # --- checker: same stack over the oracle's MoERef, on the CPU
        ref = _Stack(StableLMMLPRef).to(torch.bfloat16).to_moe(lit.MoERef, top2_2nd_expert_sampling=False)
        res = ref.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        ref.eval()
        # prefill-like, then a decode step (S = batch).  Batch 3 is avoided on purpose: the reference's
        # `len(hidden_states) == 3` test (:167) also fires for a dense layer's [3, T, M] tensor.
        for shape in ((2, 96), (4, 1)):
            x = torch.randn(*shape, M).to(torch.bfloat16)
            with torch.no_grad():
                out, losses = model(x.to(DEV))
            torch.cuda.synchronize()
            assert out.shape == x.shape and len(losses) == len(MOE_IDX)
            moe_loss = 0.01 * sum(losses)                    # :461-472
            assert moe_loss.dim() == 0 and torch.isfinite(moe_loss)
            S = shape[0] * shape[1]
            for box in feas:
                assert box["fea"].shape == (S, E) and box["fea"].dtype == torch.float32
            for idx, box in zip(MOE_IDX, feas):
                ref.layers[idx].mlp.deepspeed_moe.gate.logits_override = box["fea"]
            with torch.no_grad():
                r_out, r_losses = ref(x)
            scale = float(r_out.float().abs().max())
            err = float((out.cpu().float() - r_out.float()).abs().max())
            print(f"[f3 smoke] shape={shape} max_abs={err:.3e} ref_max={scale:.3e}")
            assert err <= 2e-2 * scale
            for a, b in zip(losses, r_losses):
                assert abs(float(a) - float(b)) <= 1e-5
            for idx in MOE_IDX:
                assert torch.equal(model.layers[idx].mlp.deepspeed_moe.exp_counts.cpu(),
                                   ref.layers[idx].mlp.deepspeed_moe.exp_counts)
    finally:
        for k in [k for k in sys.modules if k == "deepspeed" or k.startswith("deepspeed.")]:
            del sys.modules[k]
        sys.modules.update(saved_mods)


@pytest.mark.parametrize("kind", ["stablelm", "repa176"])
def test_save_and_reload_after_the_first_forward(kind, tmp_path):
    """Real checkpoints are written AFTER forwards, when every expert parameter is a view into one shared packed storage
    (and, for a RePaMLP width that is not a multiple of 64, ``down_proj.weight`` a strided view): torch.save and
    safetensors must both round-trip every key, and the reloaded block must compute the same output."""
    import moe_llava_amd
    import safetensors.torch as st
    from tests.expert_mlps import RePaMLP, StableLMMLP
    torch.manual_seed(11)
    if kind == "stablelm":
        proto = StableLMMLP(M, I)
    else:
        proto = RePaMLP(M, 200, "stablelm")
        proto.adjust_gated_ratio(0.88)                       # reparam() keeps int(200 * 0.88) = 176 channels
    kw = dict(num_experts=E, k=2, capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0,
              top2_2nd_expert_sampling=False)

    def build():
        m = moe_llava_amd.MoE(M, copy.deepcopy(proto).to(torch.bfloat16), **kw)
        if kind != "stablelm":
            for e in m.deepspeed_moe.experts.deepspeed_experts:
                e.reparam()
        return m

    moe = build()
    for p in moe.deepspeed_moe.experts.parameters():
        p.data.normal_(0, 0.05)
    moe = moe.to(DEV).eval()
    x = torch.randn(2, 50, M).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        want, _, _ = moe(x)                                  # first forward: parameters become views
    sd = moe.state_dict()
    if kind != "stablelm":                                   # the strided view itself is what the kernel reads
        assert not moe.deepspeed_moe.experts.deepspeed_experts[1].down_proj.weight.is_contiguous()
    assert all(v.is_contiguous() for v in sd.values())
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    via_torch = torch.load(buf, weights_only=True)
    path = str(tmp_path / "moe.safetensors")
    st.save_file({k: v for k, v in sd.items()}, path)
    via_st = st.load_file(path)
    for back in (via_torch, via_st):
        assert set(back) == set(sd)
        for k, v in sd.items():
            assert torch.equal(back[k].cpu(), v.cpu()), k
        fresh = build()
        res = fresh.load_state_dict(back, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        fresh = fresh.to(DEV).eval()
        with torch.no_grad():
            got, _, _ = fresh(x)
        assert torch.equal(got, want)
