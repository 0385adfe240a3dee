"""CPU tests that pin the oracle: the hand-derived known-answer case of SURVEY.md
8(a''), dense-form == index-form, and the algebraic invariants of 8(c)."""
import math

import pytest
import torch

from oracle import ds_moe_index as ix
from oracle import ds_moe_literal as lit
from oracle.expert_mlps import PhiMLPRef, QWenMLPRef, StableLMMLPRef, gelu_new


def _dense_to_index(combine):
    """Recover (expert, slot, weight) per token from the dense [S,E,C] combine."""
    S = combine.shape[0]
    out = []
    for s in range(S):
        nz = torch.nonzero(combine[s])
        out.append(sorted((int(e), int(c), float(combine[s, e, c])) for e, c in nz))
    return out


def test_known_answer_case():
    # SURVEY 8(a''): S=4,E=3,k=2,cf=0.75,min_capacity=0 => C=2
    logits = torch.tensor([[3., 2., 1.], [3., 1., 2.], [2., 3., 1.], [3., 2., 1.]])
    assert lit.capacity_of(4, 3, 0.75 * 2, 0) == 2
    l_aux, combine, dispatch, exp_counts = lit.top2gating_dense(logits.clone(), 0.75, 0)
    assert exp_counts.tolist() == [4, 3, 1]
    # SURVEY prints 1.493633; exact arithmetic on its own me/ce gives 3*(0.560113*0.75+0.311182*0.25)
    # = 1.493640 (the survey's last two digits are a rounding slip, 7e-6 off).
    assert abs(float(l_aux) - 1.4936405) < 2e-6
    assert combine.shape == (4, 3, 2)
    want = {(0, 0, 0): 0.731059, (0, 1, 1): 0.268941, (1, 0, 1): 0.731059,
            (1, 2, 0): 0.268941, (2, 1, 0): 1.0}
    for s in range(4):
        for e in range(3):
            for c in range(2):
                w = want.get((s, e, c), 0.0)
                assert abs(float(combine[s, e, c]) - w) < 2e-6, (s, e, c)
    assert float(combine[2, 1, 0]) == 1.0          # exactly 1.0 when the other choice is dropped
    assert not dispatch[3].any()                   # token 3 loses both choices

    r = ix.route_top2(logits.clone(), 2)
    assert r["expert_idx"].tolist() == [[0, 1], [0, 2], [1, 0], [0, 1]]
    assert r["slot"].tolist() == [[0, 1], [1, 0], [0, -1], [-1, -1]]
    assert r["exp_counts"].tolist() == [4, 3, 1]
    assert r["kept_counts"].tolist() == [2, 2, 1]
    assert abs(float(r["l_aux"]) - 1.4936405) < 2e-6
    offs, perm = ix.dispatch_order(r, 3)
    assert offs.tolist() == [0, 2, 4, 5]
    assert perm.tolist() == [0 * 2 + 0, 1 * 2 + 0, 2 * 2 + 0, 0 * 2 + 1, 1 * 2 + 1]


def test_capacity_formula():
    assert lit.capacity_of(704, 4, 1.5 * 2, 0) == 528
    assert lit.capacity_of(22528, 4, 1.5 * 2, 0) == 16896
    assert lit.capacity_of(704, 4, 2.0 * 2, 0) == 704           # eval default: C = S
    assert lit.capacity_of(8, 4, 1.0 * 2, 16) == 16             # min_capacity wins
    assert lit.capacity_of(1, 4, 2.0 * 2, 0) == 1               # decode step


@pytest.mark.parametrize("S,E,cf,seed", [(64, 4, 1.5, 0), (257, 4, 1.0, 1), (300, 8, 0.5, 2),
                                          (33, 3, 0.75, 3), (1, 4, 2.0, 4), (128, 4, 0.25, 5)])
@pytest.mark.parametrize("with_noise", [False, True])
def test_dense_equals_index(S, E, cf, seed, with_noise):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(S, E, generator=g)
    if seed == 2:                                    # skew: overflow expert 0
        logits[: S // 2, 0] += 2.0
    noise = lit.gumbel_noise((S, E), 99) if with_noise else None
    C = lit.capacity_of(S, E, cf * 2, 0)
    l_aux, combine, dispatch, exp_counts = lit.top2gating_dense(logits.clone(), cf, 0, True, noise)
    r = ix.route_top2(logits.clone(), C, noise)
    assert torch.equal(exp_counts, r["exp_counts"])
    assert float(l_aux) == pytest.approx(float(r["l_aux"]), abs=1e-6)
    dense = _dense_to_index(combine)
    for s in range(S):
        mine = sorted((int(r["expert_idx"][s, j]), int(r["slot"][s, j]), float(r["weight"][s, j]))
                      for j in range(2) if int(r["slot"][s, j]) >= 0 and float(r["weight"][s, j]) != 0.0)
        assert mine == dense[s], s                    # bit-identical weights, ids and slots
    # invariants
    assert int(exp_counts.sum()) == 2 * S
    assert int(r["kept_counts"].max()) <= C
    kept_per_e = dispatch.any(dim=2).sum(dim=0)
    assert kept_per_e.tolist() == r["kept_counts"].tolist()
    both = (r["slot"] >= 0).all(dim=1)
    assert torch.allclose(r["weight"][both].sum(1), torch.ones(int(both.sum())), atol=2e-7)
    one = (r["slot"] >= 0).sum(dim=1) == 1
    assert torch.equal(r["weight"][one].max(dim=1)[0], torch.ones(int(one.sum())))


def test_lowest_index_tie_break_and_uniform_laux():
    S, E = 16, 4
    logits = torch.zeros(S, E)
    r = ix.route_top2(logits, lit.capacity_of(S, E, 2.0, 0))
    assert (r["expert_idx"][:, 0] == 0).all() and (r["expert_idx"][:, 1] == 1).all()
    # uniform gates: me = 1/E; ce = one-hot on expert 0 => l_aux = E*E*mean(me*ce) = 1
    assert float(r["l_aux"]) == pytest.approx(1.0, abs=1e-6)


def test_all_tokens_one_expert_stress():
    S, E = 40, 4
    logits = torch.zeros(S, E)
    logits[:, 2] = 5.0
    logits[:, 1] = 1.0
    C = lit.capacity_of(S, E, 1.5 * 2, 0)          # 30
    r = ix.route_top2(logits, C)
    assert r["exp_counts"].tolist() == [0, S, S, 0]
    assert r["kept_counts"].tolist() == [0, C, C, 0]
    assert (r["slot"][C:] == -1).all()


def _experts(kind, M, I, E, dtype, seed=0):
    torch.manual_seed(seed)
    if kind == "stablelm":
        mk = lambda: StableLMMLPRef(M, I)
    elif kind == "phi":
        mk = lambda: PhiMLPRef(M, I)
    else:
        mk = lambda: QWenMLPRef(M, I)
    exps = [mk().to(dtype) for _ in range(E)]
    return exps


@pytest.mark.parametrize("kind", ["stablelm", "phi", "qwen"])
def test_literal_layer_equals_index_layer(kind):
    M, I, E, S = 64, 128, 4, 96
    dtype = torch.bfloat16
    torch.manual_seed(0)
    proto = _experts(kind, M, I, 1, dtype)[0]
    moe = lit.MoERef(M, proto, num_experts=E, k=2, capacity_factor=1.0, eval_capacity_factor=2.0,
                     min_capacity=0, top2_2nd_expert_sampling=False)
    # give every expert different weights
    for i, e in enumerate(moe.deepspeed_moe.experts.deepspeed_experts):
        for p in e.parameters():
            p.data = (torch.randn_like(p.float()) * 0.05).to(p.dtype)
    moe.train()
    x = (torch.randn(2, S // 2, M)).to(dtype)
    out, l_aux, exp_counts = moe(x)
    assert out.shape == x.shape and out.dtype == dtype
    assert l_aux.dim() == 0 and exp_counts.dtype == torch.int64
    logits = moe.deepspeed_moe.gate.last_logits
    C = lit.capacity_of(S, E, 1.0 * 2, 0)
    r = ix.route_top2(logits, C)
    out_ix = ix.moe_forward_index(x.reshape(-1, M), r,
                                  list(moe.deepspeed_moe.experts.deepspeed_experts)).reshape(x.shape)
    assert torch.equal(exp_counts, r["exp_counts"])
    assert float(l_aux) == pytest.approx(float(r["l_aux"]), abs=1e-6)
    err = (out.float() - out_ix.float()).abs().max()
    assert float(err) <= 1e-2 * float(out.float().abs().max()) + 1e-6
    # attribute tree / tags
    assert moe.expert_group_name == "ep_size_1"
    for p in moe.deepspeed_moe.experts.parameters():
        assert p.allreduce is False and p.group_name == "ep_size_1"
    keys = list(moe.state_dict().keys())
    assert "deepspeed_moe.gate.wg.weight" in keys
    assert any(k.startswith("deepspeed_moe.experts.deepspeed_experts.3.") for k in keys)


def test_top1_rts_dense_equals_index():
    S, E = 64, 4
    torch.manual_seed(3)
    logits = torch.randn(S, E)
    logits[:, 1] += 1.0
    u = torch.rand(S, E)
    C = lit.capacity_of(S, E, 1.0, 4)
    l_aux, combine, dispatch, exp_counts = lit.top1gating_dense(logits, 1.0, 4, rts_uniform=u)
    r = ix.route_top1_rts(logits, C, u)
    assert torch.equal(exp_counts, r["exp_counts"])
    assert float(l_aux) == pytest.approx(float(r["l_aux"]), abs=1e-6)
    dense = _dense_to_index(combine)
    for s in range(S):
        mine = [(int(r["expert_idx"][s, 0]), int(r["slot"][s, 0]), float(r["weight"][s, 0]))] \
            if int(r["slot"][s, 0]) >= 0 else []
        assert mine == dense[s]


@pytest.mark.parametrize("use_rts,masked,rsample", [(True, True, False), (False, False, False), (False, True, True),
                                                    (True, False, True)])
def test_top1_variants_dense_equals_index(use_rts, masked, rsample):
    """used_token, the no-RTS rule (first C member tokens) and 'RSample' noise: dense restatement == index form."""
    S, E = 96, 4
    torch.manual_seed(11)
    logits = torch.randn(S, E)
    logits[:, 2] += 1.5
    u = torch.rand(S, E) if use_rts else None
    used = (torch.rand(S) > 0.25).long() if masked else None
    noise = lit.gumbel_noise((S, E), 5) if rsample else None
    C = lit.capacity_of(S, E, 0.75, 0)
    l_aux, combine, dispatch, exp_counts = lit.top1gating_dense(logits, 0.75, 0, used, True, use_rts, u, noise)
    r = ix.route_top1_rts(logits, C, u, used, noise)
    assert torch.equal(exp_counts, r["exp_counts"])
    assert int(r["kept_counts"].sum()) < int(r["exp_counts"].sum())          # something was dropped
    assert float(l_aux) == pytest.approx(float(r["l_aux"]), abs=1e-6)
    dense = _dense_to_index(combine)
    for s in range(S):
        mine = [(int(r["expert_idx"][s, 0]), int(r["slot"][s, 0]), float(r["weight"][s, 0]))] \
            if int(r["slot"][s, 0]) >= 0 else []
        assert mine == dense[s]


def test_gelu_new_matches_transformers_if_present():
    try:
        from transformers.activations import ACT2FN
    except Exception:
        pytest.skip("transformers not importable")
    x = torch.linspace(-6, 6, 1001)
    assert torch.allclose(gelu_new(x), ACT2FN["gelu_new"](x), atol=1e-6)
    assert math.isclose(float(gelu_new(torch.tensor(0.0))), 0.0)
