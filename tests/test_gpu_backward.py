"""GPU parity of the backward pass: gradients of the HIP block against torch autograd over the
literal dense-mask oracle on the CPU (bf16 experts, fp32 gate), with the kernel's own fp32 logits injected so that
both sides take the same routing decisions.  Loss = <out, R> + c * l_aux, the shape of the reference's training
loss (cross-entropy through the block output plus router_aux_loss_coef * l_aux, llava_stablelm_moe.py:461-472).

Tolerance: gradients are bf16 tensors produced by bf16 GEMMs on both sides with different accumulation orders, so
they are compared relative to the tensor's scale: max|d| <= 2e-2 * max|ref| (+ the norm-wise error printed)."""
import pytest
import torch

from oracle import ds_moe_literal as lit
from tests.test_gpu_parity import DEV, _build_pair

pytestmark = pytest.mark.gpu


def _cmp(name, got, want, tol=2e-2):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    nrm = float((got - want).norm() / max(float(want.norm()), 1e-30))
    print(f"[bwd parity] {name}: max_abs={err:.3e} ref_max={scale:.3e} rel_to_scale={err / max(scale, 1e-30):.3e} "
          f"rel_l2={nrm:.3e}")
    assert err <= tol * scale + 1e-6, name
    assert nrm <= tol, name


def _run_both(kind, M, I, E, shape, cf, noisy, seed=11, aux_coef=0.3, use_residual=False, drop_tokens=True, skew=0.0):
    ref, mine, x = _build_pair(kind, M, I, E, shape, cf, seed=seed, noisy=noisy, use_residual=use_residual,
                               drop_tokens=drop_tokens)
    if skew:                                         # push the tokens towards expert 0: its capacity share overflows
        for m in (ref, mine):
            m.deepspeed_moe.gate.wg.weight.data[0] += skew * torch.sign(m.deepspeed_moe.gate.wg.weight.data[0])
        x = (x.float().abs() * torch.sign(ref.deepspeed_moe.gate.wg.weight.data[0])).to(x.dtype)
    S = x.numel() // M
    if noisy:
        noise = lit.gumbel_noise((S, E), 99)
        mine.deepspeed_moe.gate.noise_override = noise.to(DEV)
        ref.deepspeed_moe.gate.noise = noise
    g = torch.Generator().manual_seed(seed + 1)
    R = torch.randn(x.shape, generator=g).to(torch.bfloat16)
    xg = x.to(DEV).requires_grad_(True)
    out, l_aux, _ = mine(xg)
    assert out.requires_grad and l_aux.requires_grad
    loss = (out.float() * R.to(DEV).float()).sum() + aux_coef * l_aux
    loss.backward()
    torch.cuda.synchronize()
    ref.deepspeed_moe.gate.logits_override = mine.deepspeed_moe.gate.last_logits.cpu()
    xr = x.clone().requires_grad_(True)
    r_out, r_laux, _ = ref(xr)
    r_loss = (r_out.float() * R.float()).sum() + aux_coef * r_laux
    r_loss.backward()
    return ref, mine, xg, xr, out, r_out


@pytest.mark.parametrize("kind,M,I,E,shape,cf", [
    ("stablelm", 64, 128, 4, (2, 48), 1.0),
    ("stablelm", 128, 320, 4, (3, 111), 1.5),
    ("qwen", 128, 192, 4, (1, 300), 0.5),
    ("qwen_bias", 64, 64, 2, (1, 70), 1.5),
    ("phi", 128, 256, 4, (2, 130), 1.5),
    ("phi", 64, 576, 8, (1, 257), 1.0),
    # several 256-wide tiles in every GEMM of the step (weight gradients: 2 x 3 and 6 x 2 tiles per expert)
    ("stablelm", 512, 768, 4, (2, 333), 1.5),
    ("phi", 320, 1024, 2, (1, 700), 1.0),
    # more than 16 local experts: the weight-gradient tile order keeps the experts in index order (no ranking by rows)
    ("stablelm", 64, 128, 24, (2, 190), 2.0),
])
@pytest.mark.parametrize("noisy", [False, True])
def test_block_gradients_vs_oracle_autograd(kind, M, I, E, shape, cf, noisy):
    ref, mine, xg, xr, out, r_out = _run_both(kind, M, I, E, shape, cf, noisy)
    tag = f"{kind} M{M} I{I} E{E} S{xg.numel() // M} noisy={noisy}"
    _cmp(tag + " out", out, r_out, tol=1e-2)
    _cmp(tag + " dx", xg.grad, xr.grad)
    _cmp(tag + " dwg", mine.deepspeed_moe.gate.wg.weight.grad, ref.deepspeed_moe.gate.wg.weight.grad)
    rp = dict(ref.deepspeed_moe.experts.named_parameters())
    n = 0
    for name, p in mine.deepspeed_moe.experts.named_parameters():
        assert p.grad is not None and p.grad.dtype == p.dtype and p.grad.shape == p.shape, name
        _cmp(f"{tag} {name}", p.grad, rp[name].grad)
        n += 1
    assert n == len(rp)


@pytest.mark.parametrize("kind,M,I,E,shape,cf", [("stablelm", 128, 320, 4, (3, 111), 1.5), ("phi", 64, 256, 8, (1, 257), 1.0),
                                              ("qwen", 2048, 128, 4, (1, 70), 2.0)])
def test_jitter_policy_gradients_vs_oracle_autograd(kind, M, I, E, shape, cf):
    """noisy_gate_policy='Jitter' in training: forward and every gradient against autograd over the literal oracle that is
    given the SAME factors (read back from the seed with ops.gate_jitter) -- the gate input is x.float() * factors, so dwg
    sees the jittered input and the gate's share of dx is scaled by the factors."""
    from moe_llava_amd import ops
    ref, mine, x = _build_pair(kind, M, I, E, shape, cf, seed=21, noisy=False)
    S = x.numel() // M
    for m in (ref, mine):
        m.deepspeed_moe.gate.noisy_gate_policy = "Jitter"
    seed = 0x0bad_cafe_1234_5677
    mine.deepspeed_moe.gate.jitter_seed_override = seed
    ref.deepspeed_moe.gate.jitter = ops.gate_jitter(seed, S, M, DEV).cpu()
    g = torch.Generator().manual_seed(22)
    R = torch.randn(x.shape, generator=g).to(torch.bfloat16)
    xg = x.to(DEV).requires_grad_(True)
    out, l_aux, _ = mine(xg)
    assert mine.deepspeed_moe.gate.last_jitter_seed == seed
    ((out.float() * R.to(DEV).float()).sum() + 0.3 * l_aux).backward()
    torch.cuda.synchronize()
    lg_mine = mine.deepspeed_moe.gate.last_logits.cpu()
    xr = x.clone().requires_grad_(True)
    # first the oracle's own jittered GEMV against the kernel's logits, then the kernel's logits injected for the routing
    r_out, r_laux, _ = ref(xr)
    lg_ref = ref.deepspeed_moe.gate.last_logits
    assert float((lg_mine - lg_ref).abs().max()) <= 1e-4 * max(1.0, float(lg_ref.abs().max()))
    ref.deepspeed_moe.gate.logits_override = lg_mine
    xr = x.clone().requires_grad_(True)
    r_out, r_laux, _ = ref(xr)
    ((r_out.float() * R.float()).sum() + 0.3 * r_laux).backward()
    tag = f"jitter {kind} M{M}"
    _cmp(tag + " out", out, r_out, tol=1e-2)
    _cmp(tag + " dx", xg.grad, xr.grad)
    _cmp(tag + " dwg", mine.deepspeed_moe.gate.wg.weight.grad, ref.deepspeed_moe.gate.wg.weight.grad)
    rp = dict(ref.deepspeed_moe.experts.named_parameters())
    for name, p in mine.deepspeed_moe.experts.named_parameters():
        _cmp(f"{tag} {name}", p.grad, rp[name].grad)


@pytest.mark.parametrize("kind,M,I,E,shape", [("stablelm", 128, 320, 4, (2, 90)), ("phi", 64, 128, 2, (1, 150))])
def test_residual_moe_forward_and_gradients_vs_oracle(kind, M, I, E, shape):
    """`use_residual=True` (`MoE.forward`'s blend `out * c[..., :1] + mlp(x) * c[..., 1:]`, c =
    softmax(coefficient(x))): the dense branch and the blend run as the reference's own torch modules around the HIP
    block; output and every gradient (x through all three branches, `coefficient`, `mlp`, gate, experts) against the
    oracle."""
    ref, mine, xg, xr, out, r_out = _run_both(kind, M, I, E, shape, 1.5, False, seed=21, use_residual=True)
    assert mine.mlp is not mine.deepspeed_moe.experts.deepspeed_experts[0]
    tag = f"residual {kind}"
    _cmp(tag + " out", out, r_out, tol=1e-2)
    _cmp(tag + " dx", xg.grad, xr.grad)
    rp = dict(ref.named_parameters())
    seen = set()
    for name, p in mine.named_parameters():
        assert p.grad is not None, name
        # coefficient.* are torch's own bf16 gradients on both sides (GPU reduction order vs CPU): two bias elements that
        # are sums of a few hundred cancelling bf16 terms differ by several ulp from run to run
        _cmp(f"{tag} {name}", p.grad, rp[name].grad, tol=8e-2 if name.startswith("coefficient") else 2e-2)
        seen.add(name.split(".")[0])
    assert {"coefficient", "mlp", "deepspeed_moe"} <= seen


@pytest.mark.parametrize("kind,M,I,E,shape,noisy", [("stablelm", 128, 192, 4, (2, 150), False), ("phi", 64, 128, 8, (1, 333), True)])
def test_drop_tokens_false_forward_and_gradients_vs_oracle(kind, M, I, E, shape, noisy):
    """`drop_tokens=False`: the capacity becomes the busiest expert's count, nobody is dropped, although
    the capacity factor alone (0.25) would drop most of expert 0's tokens.  Block output and gradients against the oracle."""
    ref, mine, xg, xr, out, r_out = _run_both(kind, M, I, E, shape, 0.25, noisy, seed=33, drop_tokens=False, skew=0.3)
    r = mine.deepspeed_moe.last_routing
    S = xg.numel() // M
    assert int((r.slot.cpu() < 0).sum()) == 0 and int(r.kept_counts.cpu().sum()) == 2 * S
    assert int(r.exp_counts.cpu().max()) > -(-S // E)            # an expert holds more than an even share
    tag = f"no-drop {kind}"
    _cmp(tag + " out", out, r_out, tol=1e-2)
    _cmp(tag + " dx", xg.grad, xr.grad)
    rp = dict(ref.named_parameters())
    for name, p in mine.named_parameters():
        _cmp(f"{tag} {name}", p.grad, rp[name].grad)


def test_training_forward_equals_inference_forward():
    """The training path (plain fc1 + activation kernel, pre-activations kept) gives bit-identical outputs to the
    fused-epilogue inference path."""
    for kind in ("stablelm", "phi"):
        _, mine, x = _build_pair(kind, 128, 256, 4, (2, 200), 1.0, seed=5)
        xd = x.to(DEV)
        with torch.no_grad():
            o1, l1, _ = mine(xd)
        o2, l2, _ = mine(xd.clone().requires_grad_(True))
        assert torch.equal(o1, o2.detach()) and float(l1) == float(l2.detach())


def test_frozen_experts_and_idle_expert():
    """Only wg trainable (stage of MoE-tuning that trains the router) and an expert that receives no token."""
    ref, mine, x = _build_pair("stablelm", 64, 128, 4, (1, 64), 1.0, seed=9)
    for p in mine.deepspeed_moe.experts.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        mine.deepspeed_moe.gate.wg.weight[3] = -5.0 * mine.deepspeed_moe.gate.wg.weight[0].abs() - 1.0   # starve expert 3
    out, l_aux, cnt = mine(x.abs().to(DEV))
    (out.float().sum() + l_aux).backward()
    torch.cuda.synchronize()
    assert mine.deepspeed_moe.gate.wg.weight.grad is not None
    assert all(p.grad is None for p in mine.deepspeed_moe.experts.parameters())
    # now with trainable experts: the starved expert's weight gradients are exactly zero
    for p in mine.deepspeed_moe.experts.parameters():
        p.requires_grad_(True)
    out, l_aux, cnt = mine(x.abs().to(DEV))
    (out.float().sum() + l_aux).backward()
    torch.cuda.synchronize()
    if int(cnt[3]) == 0:
        for p in mine.deepspeed_moe.experts.deepspeed_experts[3].parameters():
            assert float(p.grad.float().abs().max()) == 0.0
    for p in mine.deepspeed_moe.experts.deepspeed_experts[0].parameters():
        assert torch.isfinite(p.grad.float()).all()


def test_cfg1_shape_backward_finite_and_linear():
    """Config-1 block (S=704, M=2048, I=5632): gradients are finite and linear in the upstream gradient."""
    import moe_llava_amd
    from tests.expert_mlps import StableLMMLP
    torch.manual_seed(0)
    M, I, E, S = 2048, 5632, 4, 704
    moe = moe_llava_amd.MoE(M, StableLMMLP(M, I), num_experts=E, k=2, capacity_factor=1.5, min_capacity=0,
                            top2_2nd_expert_sampling=False).to(torch.bfloat16).to(DEV).train()
    for p in moe.deepspeed_moe.experts.parameters():
        p.data.normal_(0, 0.02)
    moe.deepspeed_moe.gate.wg.weight.data = moe.deepspeed_moe.gate.wg.weight.data.float()
    x = torch.randn(1, S, M, device=DEV).to(torch.bfloat16)
    R = torch.randn(1, S, M, device=DEV).to(torch.bfloat16)
    grads = []
    for scale in (1.0, 2.0):
        moe.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        out, l_aux, _ = moe(xg)
        (out.float() * (scale * R.float())).sum().backward()
        torch.cuda.synchronize()
        grads.append([xg.grad.float()] + [p.grad.float() for p in moe.parameters()])
    for a, b in zip(*grads):
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
        assert float(a.abs().max()) > 0
        # power-of-two scaling of the upstream gradient is exact in bf16/fp32
        assert torch.equal(2.0 * a, b)


def test_cfg2_full_size_backward_slices_vs_fp32_reference():
    """BASELINE config 2 at its real size (S = 22 528, M = 2048, I = 5632, 4 experts, cf 1.5): one training step of the HIP
    block, then on the CPU, from the step's inputs alone (x, weights, the kernel's routing, the upstream gradient R):
      * for every expert, 48 columns of the hidden width (both sides of a 256-column tile edge and the last columns):
        the down_proj weight-gradient columns dW2[:, c] = dY_e^T h[:, c], and the gate_proj / up_proj weight-gradient rows
        dW1[c, :] = dAB[:, c]^T X_e, with dY, h, dH, dAB rebuilt in fp32 torch with the bf16 rounding points of autograd
        over the reference's bf16 modules;
      * dx of ~100 tokens (random ones and the ones whose compact rows sit at expert ends and 256-row tile edges) by
        torch autograd over the oracle expert modules for those tokens, the gate included.
    The small-shape tests compare every gradient element; this one is the check that nothing breaks at real K, real row
    counts (176 row tiles, ragged expert ends) and real tile counts."""
    import moe_llava_amd
    from oracle.expert_mlps import StableLMMLPRef
    from tests._parity import edge_tokens
    from tests.expert_mlps import StableLMMLP
    torch.manual_seed(5)
    M, I, E, B, T = 2048, 5632, 4, 32, 704
    S = B * T
    moe = moe_llava_amd.MoE(M, StableLMMLP(M, I), num_experts=E, k=2, capacity_factor=1.5, min_capacity=0,
                            top2_2nd_expert_sampling=False).to(torch.bfloat16)
    for p in moe.deepspeed_moe.experts.parameters():
        p.data.normal_(0, 0.02)
    moe.deepspeed_moe.gate.wg.weight.data = torch.randn(E, M) * 0.05
    moe = moe.to(DEV).train()
    x = torch.randn(B, T, M, device=DEV)
    x[:, :576] += 0.5                                          # image rows skew the router: drops happen
    x = x.to(torch.bfloat16)
    R = torch.randn(B, T, M, device=DEV).to(torch.bfloat16)
    aux_c = 0.01
    xg = x.clone().requires_grad_(True)
    out, l_aux, _ = moe(xg)
    torch.autograd.backward([out, l_aux], [R, torch.tensor(aux_c, device=DEV)])
    torch.cuda.synchronize()
    r = moe.deepspeed_moe.last_routing
    offs = r.expert_offsets.cpu().tolist()
    perm = r.perm.cpu().long()
    idx, slot, wt = r.expert_idx.cpu().long(), r.slot.cpu().long(), r.weight.cpu()
    assert int((slot < 0).sum()) > 0 and offs[-1] > 40000
    xc, Rc = x.reshape(S, M).cpu(), R.reshape(S, M).cpu()
    rb = lambda t: t.to(torch.bfloat16).float()                # a bf16 rounding point
    cols = torch.tensor(list(range(0, 16)) + list(range(248, 264)) + list(range(I - 16, I)))
    experts = list(moe.deepspeed_moe.experts.deepspeed_experts)
    for e in range(E):
        rows = perm[offs[e]:offs[e + 1]]
        tok, ch = rows // 2, rows % 2
        Xe = xc[tok].float()
        wq = rb(wt[tok, ch])                                   # combine weights are rounded to bf16 first
        dY = rb(wq[:, None] * Rc[tok].float())                 # combine backward: dY[p] = bf16(w) * dOut[token]
        mod = experts[e]
        Wg, Wu, Wd = (mod.gate_proj.weight.detach().cpu().float(), mod.up_proj.weight.detach().cpu().float(),
                      mod.down_proj.weight.detach().cpu().float())
        a, b = rb(Xe @ Wg[cols].T), rb(Xe @ Wu[cols].T)
        sig = torch.sigmoid(a)
        sg = rb(a * sig)
        h = rb(sg * b)
        _cmp(f"cfg2 expert {e} down_proj.weight.grad[:, cols]", mod.down_proj.weight.grad[:, cols.to(DEV)], dY.T @ h)
        dH = rb(dY @ Wd[:, cols])
        dsg, db = rb(dH * b), rb(dH * sg)
        da = rb(dsg * (sig * (1 + a * (1 - sig))))
        _cmp(f"cfg2 expert {e} gate_proj.weight.grad[cols]", mod.gate_proj.weight.grad[cols.to(DEV)], da.T @ Xe)
        _cmp(f"cfg2 expert {e} up_proj.weight.grad[cols]", mod.up_proj.weight.grad[cols.to(DEV)], db.T @ Xe)
    # ---- dx of sampled tokens: autograd over the oracle modules, routing decisions taken from the kernel ----
    smp = torch.unique(torch.cat([torch.randperm(S, generator=torch.Generator().manual_seed(3))[:48],
                                  edge_tokens(offs, r.perm, 2, 256, seed=4, per_expert=2)]))
    cpu_experts = [StableLMMLPRef(M, I).to(torch.bfloat16) for _ in range(E)]
    for ce_, ge in zip(cpu_experts, experts):
        ce_.load_state_dict({k: v.detach().cpu() for k, v in ge.state_dict().items()})
    wg = moe.deepspeed_moe.gate.wg.weight.detach().cpu().float()
    n1 = torch.bincount(idx[:, 0], minlength=E).float()
    xs = xc[smp].clone().requires_grad_(True)
    gates = torch.softmax(xs.float() @ wg.T, dim=1)
    keep = (slot[smp] >= 0).float()
    g = gates.gather(1, idx[smp]) * keep
    w = g / torch.clamp(g.sum(1, keepdim=True), min=torch.finfo(torch.float32).eps)
    acc = torch.zeros(smp.numel(), M)
    for e in range(E):
        for j in range(2):
            on = torch.nonzero((idx[smp, j] == e) & (slot[smp, j] >= 0))[:, 0]
            if on.numel():
                y = cpu_experts[e](xs[on])
                acc = acc.index_add(0, on, w[on, j, None].to(torch.bfloat16).float() * y.float())
    o = acc.to(torch.bfloat16)
    loss = (o.float() * Rc[smp].float()).sum() + aux_c * (E / S) * (gates * (n1 / S)).sum()   # l_aux = E * sum_e me_e ce_e
    loss.backward()
    _cmp("cfg2 out (sampled tokens)", out.reshape(S, M)[smp.to(DEV)], o, tol=1e-2)
    _cmp(f"cfg2 dx ({smp.numel()} sampled tokens)", xg.grad.reshape(S, M)[smp.to(DEV)], xs.grad)


@pytest.mark.parametrize("reentrant", [True, False])
def test_activation_checkpointing_recompute_matches(reentrant):
    """The reference fine-tunes with --gradient_checkpointing True (scripts/v1/stablelm/finetune_moe.sh:44): every decoder
    layer runs under torch.utils.checkpoint, so the block's forward is re-run inside backward.  The recomputation must take
    the same routing decisions (the 2nd-expert Gumbel seed comes from torch's CPU generator, whose state checkpoint restores)
    and give the gradients of the plain run, bit for bit."""
    from torch.utils.checkpoint import checkpoint
    import moe_llava_amd
    from tests.expert_mlps import StableLMMLP
    M, I, E = 128, 256, 4
    torch.manual_seed(3)
    moe = moe_llava_amd.MoE(M, StableLMMLP(M, I), num_experts=E, k=2, capacity_factor=1.0, min_capacity=0)
    moe = moe.to(torch.bfloat16).to(DEV).train()           # top2_2nd_expert_sampling stays on (upstream default): noisy routing
    for p in moe.deepspeed_moe.experts.parameters():
        p.data.normal_(0, 0.05)
    x = torch.randn(2, 150, M, device=DEV).to(torch.bfloat16)
    R = torch.randn(2, 150, M, device=DEV).to(torch.bfloat16)

    def layer(h):                                          # the decoder glue of llava_stablelm_moe.py:163-170, reduced
        out, l_aux, _ = moe(h)
        return h + out, l_aux

    def run(ckpt):
        moe.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        torch.manual_seed(77)                              # same generator state in front of both runs
        if ckpt:
            y, l_aux = checkpoint(layer, xg, use_reentrant=reentrant)
        else:
            y, l_aux = layer(xg)
        ((y.float() * R.float()).sum() + 0.01 * l_aux).backward()
        torch.cuda.synchronize()
        return [y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in moe.parameters()]

    plain, ck = run(False), run(True)
    assert len(plain) == len(ck)
    for a, b in zip(plain, ck):
        assert torch.equal(a, b)
