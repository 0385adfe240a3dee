"""GPU parity of top-1 gating: routing bit-exact against the index oracle given the same fp32
logits and the same injected random numbers (RTS uniforms, RSample Gumbel noise); whole block (k=1) forward and
backward against the literal dense-mask oracle; in-kernel random numbers checked through their invariants."""
import pytest
import torch

from oracle import ds_moe_index as ix
from oracle import ds_moe_literal as lit
from tests._parity import assert_out_close
from tests.test_gpu_parity import DEV, _mk_experts, _report

pytestmark = pytest.mark.gpu


def _check(logits, C, u=None, used=None, noise=None, drop=True):
    from moe_llava_amd import ops
    S, E = logits.shape
    r = ops.routing_top1(logits.to(DEV), C, None if used is None else used.to(DEV), u is not None,
                         None if u is None else u.to(DEV), 0, None if noise is None else noise.to(DEV), 0, drop)
    torch.cuda.synchronize()
    want = ix.route_top1_rts(logits, C, u, used, noise, drop)
    offs, perm = ix.dispatch_order(want, E)
    assert torch.equal(r.expert_idx.cpu(), want["expert_idx"])
    assert torch.equal(r.slot.cpu(), want["slot"])
    assert torch.equal(r.exp_counts.cpu(), want["exp_counts"])
    assert torch.equal(r.kept_counts.cpu(), want["kept_counts"])
    assert torch.equal(r.expert_offsets.cpu(), offs)
    n = int(offs[-1])
    assert torch.equal(r.perm.cpu()[:n], perm)
    assert torch.allclose(r.weight.cpu(), want["weight"], rtol=0, atol=1e-6)
    assert abs(float(r.l_aux) - float(want["l_aux"])) <= 1e-5
    return r, want


@pytest.mark.parametrize("S,E,cf,seed", [(64, 4, 1.0, 0), (704, 4, 0.5, 1), (3000, 8, 0.75, 2), (22528, 4, 1.0, 3),
                                         (9000, 16, 0.25, 4), (257, 64, 1.0, 5), (131072, 8, 0.9, 6), (5, 2, 1.0, 7)])
@pytest.mark.parametrize("rts", [True, False])
def test_top1_routing_bit_exact(S, E, cf, seed, rts):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(S, E, generator=g)
    logits[:, 0] += 0.8                                     # expert 0 overflows
    u = torch.rand(S, E, generator=g) if rts else None
    C = lit.capacity_of(S, E, cf, 0)
    r, want = _check(logits, C, u)
    assert int(want["kept_counts"][0]) == min(C, int(want["exp_counts"][0]))


def test_top1_used_token_rsample_nodrop_and_ties():
    g = torch.Generator().manual_seed(42)
    S, E = 2000, 4
    logits = torch.randn(S, E, generator=g)
    logits[:, 3] += 1.0
    u = torch.rand(S, E, generator=g)
    used = (torch.rand(S, generator=g) > 0.3).long()
    noise = lit.gumbel_noise((S, E), 9)
    C = lit.capacity_of(S, E, 0.5, 4)
    _check(logits, C, u, used)
    _check(logits, C, u, used, noise)
    _check(logits, C, None, used, noise)
    _check(logits, C, u, None, None, drop=False)
    # heavy ties in the uniforms: only 4 distinct values -> the lower token index wins inside a value
    ut = (torch.randint(0, 4, (S, E), generator=g).float()) / 4
    _check(logits, C, ut)
    _check(logits, 0, u)                                     # capacity 0: nothing kept
    _check(torch.zeros(S, E), C, u)                          # all logits equal: every token picks expert 0


def test_top1_routing_fuzz_vs_index_oracle():
    """80 seeded random top-1 configurations against the index oracle, bit-exact: sizes around the kernels' boundaries, odd
    expert counts, RTS on / off (coarse uniforms give ties), `used_token` masks, RSample noise, starving capacities,
    drop_tokens off."""
    import random
    rnd = random.Random(777)
    edges = [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049, 4097]
    for case in range(80):
        S = rnd.choice(edges) if rnd.random() < 0.5 else rnd.randint(1, 7000)
        E = rnd.choice([2, 3, 4, 4, 5, 8, 8, 13, 16, 64])
        g = torch.Generator().manual_seed(5000 + case)
        logits = torch.randn(S, E, generator=g) * rnd.choice([0.2, 1.0, 1.0, 5.0])
        if rnd.random() < 0.4:
            logits[:, rnd.randrange(E)] += 1.5
        if rnd.random() < 0.2:
            logits = logits.round()
        u = None
        if rnd.random() < 0.6:
            u = torch.rand(S, E, generator=g)
            if rnd.random() < 0.3:
                u = (u * 8).floor() / 8
        used = (torch.rand(S, generator=g) > 0.25).long() if rnd.random() < 0.4 else None
        noise = lit.gumbel_noise((S, E), 31 + case) if rnd.random() < 0.3 else None
        drop = rnd.random() < 0.85
        cf = rnd.choice([0.05, 0.25, 0.5, 1.0, 1.5, 3.0])
        C = lit.capacity_of(S, E, cf, rnd.choice([0, 0, 4, 9])) if drop else 0
        try:
            _check(logits, C, u, used, noise, drop)
        except AssertionError as e:
            raise AssertionError(f"case {case}: S={S} E={E} cf={cf} C={C} drop={drop} rts={u is not None} "
                                 f"used={used is not None} noisy={noise is not None}: {e}")


def test_top1_in_kernel_random_numbers():
    from moe_llava_amd import ops
    S, E = 8192, 4
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(S, E, generator=g)
    logits[:, 1] += 1.0
    C = lit.capacity_of(S, E, 0.5, 0)
    a = ops.routing_top1(logits.to(DEV), C, None, True, None, 1234)
    b = ops.routing_top1(logits.to(DEV), C, None, True, None, 1234)
    c = ops.routing_top1(logits.to(DEV), C, None, True, None, 99)
    first = ops.routing_top1(logits.to(DEV), C, None, False)
    torch.cuda.synchronize()
    assert torch.equal(a.slot, b.slot)                                       # same seed, same selection
    assert not torch.equal(a.slot, c.slot)                                   # other seed, other selection
    n1 = a.exp_counts.cpu()
    assert torch.equal(a.kept_counts.cpu().long(), torch.clamp(n1, max=C))
    # the kept set of the overflowing expert is spread over the whole sequence, not its head (what RTS is for)
    tok = torch.nonzero((a.expert_idx[:, 0] == 1) & (a.slot[:, 0] >= 0))[:, 0].float()
    tok_first = torch.nonzero((first.expert_idx[:, 0] == 1) & (first.slot[:, 0] >= 0))[:, 0].float()
    assert float(tok.mean()) > 1.5 * float(tok_first.mean())
    # RSample: different seeds move some first choices
    d = ops.routing_top1(logits.to(DEV), C, None, False, None, 0, None, 7)
    e = ops.routing_top1(logits.to(DEV), C, None, False, None, 0, None, 8)
    assert not torch.equal(d.expert_idx, e.expert_idx)


@pytest.mark.parametrize("kind,M,I,E,shape,cf", [("stablelm", 128, 320, 4, (2, 150), 0.75), ("phi", 64, 128, 8, (1, 257), 1.0)])
@pytest.mark.parametrize("policy", [None, "RSample"])
def test_top1_block_forward_backward_vs_literal_oracle(kind, M, I, E, shape, cf, policy):
    import moe_llava_amd
    exps = _mk_experts(kind, M, I, E, 13)
    kw = dict(num_experts=E, k=1, capacity_factor=cf, eval_capacity_factor=2.0, min_capacity=0, noisy_gate_policy=policy)
    ref = lit.MoERef(M, exps[0], **kw)
    for dst, src in zip(ref.deepspeed_moe.experts.deepspeed_experts, exps):
        dst.load_state_dict(src.state_dict())
    g = torch.Generator().manual_seed(113)
    ref.deepspeed_moe.gate.wg.weight.data = torch.randn(E, M, generator=g) * 0.1
    mine = moe_llava_amd.MoE(M, exps[0], **kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.to(DEV).train()
    ref.train()
    x = torch.randn(*shape, M, generator=g).to(torch.bfloat16)
    x[..., : M // 2] += 0.3
    R = torch.randn(*shape, M, generator=g).to(torch.bfloat16)
    S = x.numel() // M
    u = torch.rand(S, E, generator=g)
    used = (torch.rand(S, generator=g) > 0.2).long()
    mine.deepspeed_moe.gate.rts_uniform_override = u.to(DEV)
    ref.deepspeed_moe.gate.rts_uniform = u
    if policy == "RSample":
        noise = lit.gumbel_noise((S, E), 3)
        mine.deepspeed_moe.gate.noise_override = noise.to(DEV)
        ref.deepspeed_moe.gate.noise = noise
    xg = x.to(DEV).requires_grad_(True)
    out, l_aux, cnt = mine(xg, used.to(DEV))
    ((out.float() * R.to(DEV).float()).sum() + 0.3 * l_aux).backward()
    torch.cuda.synchronize()
    r = mine.deepspeed_moe.last_routing
    assert r.top_k == 1 and int((r.slot < 0).sum()) > 0
    ref.deepspeed_moe.gate.logits_override = mine.deepspeed_moe.gate.last_logits.cpu()
    xr = x.clone().requires_grad_(True)
    r_out, r_laux, r_cnt = ref(xr, used)
    ((r_out.float() * R.float()).sum() + 0.3 * r_laux).backward()
    assert torch.equal(cnt.cpu(), r_cnt)
    assert abs(float(l_aux.detach()) - float(r_laux.detach())) <= 1e-5
    tag = f"top1 {kind} policy={policy}"
    assert_out_close(tag + " out", out.detach().cpu(), r_out.detach())
    for name, a, b in [("dx", xg.grad, xr.grad),
                       ("dwg", mine.deepspeed_moe.gate.wg.weight.grad, ref.deepspeed_moe.gate.wg.weight.grad)]:
        err, scale = _report(f"{tag} {name}", a.cpu(), b)
        assert err <= 2e-2 * scale + 1e-6
    rp = dict(ref.deepspeed_moe.experts.named_parameters())
    for n, p in mine.deepspeed_moe.experts.named_parameters():
        err, scale = _report(f"{tag} {n}", p.grad.cpu(), rp[n].grad)
        assert err <= 2e-2 * scale + 1e-6
    # inference path, eval capacity (2.0: nothing dropped except unused tokens)
    mine.eval(); ref.eval()
    with torch.no_grad():
        o2, _, c2 = mine(x.to(DEV), used.to(DEV))
        ref.deepspeed_moe.gate.logits_override = mine.deepspeed_moe.gate.last_logits.cpu()
        r2, _, rc2 = ref(x, used)
    assert torch.equal(c2.cpu(), rc2)
    assert_out_close(tag + " eval out", o2.cpu(), r2)
    if policy == "RSample":
        # eval routing is noise-free: upstream hands top1gating `noisy_gate_policy if self.training else None`
        lg = mine.deepspeed_moe.gate.last_logits
        keep = used.to(DEV) != 0
        first = mine.deepspeed_moe.last_routing.expert_idx[:, 0].long()
        assert torch.equal(first[keep], torch.softmax(lg, 1).argmax(1)[keep])
