"""GPU parity tests: the HIP path (through the C ABI, via moe_llava_amd.ops / MoE) against the CPU oracle.

Bars (BASELINE.json north_star): expert ids, slots, drops, exp_counts, kept_counts, offsets, perm BIT-EXACT given
fp32 logits and lowest-index tie-break; l_aux within 1e-5; outputs within bf16 tolerance rtol 1e-2 (max-abs
printed).  The oracle is "parity unpinned" against upstream DeepSpeed (see oracle/__init__.py).
"""
import pytest
import torch

from oracle import ds_moe_index as ix
from oracle import ds_moe_literal as lit
from oracle.expert_mlps import PhiMLPRef, QWenMLPRef, StableLMMLPRef
from tests._parity import assert_out_close, edge_tokens, knob, report as _report

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from moe_llava_amd import ops
    return ops


def _check_routing(logits_cpu, C, noise_cpu=None, drop=True):
    ops = _ops()
    S, E = logits_cpu.shape
    ref = ix.route_top2(logits_cpu.clone(), C, noise_cpu, drop)
    r = ops.routing_top2(logits_cpu.to(DEV), C, None if noise_cpu is None else noise_cpu.to(DEV), drop)
    torch.cuda.synchronize()
    assert torch.equal(r.expert_idx.cpu(), ref["expert_idx"]), "expert ids differ"
    assert torch.equal(r.slot.cpu(), ref["slot"]), "slots / drops differ"
    assert torch.equal(r.exp_counts.cpu(), ref["exp_counts"]), "exp_counts differ"
    assert torch.equal(r.kept_counts.cpu(), ref["kept_counts"]), "kept_counts differ"
    offs, perm = ix.dispatch_order(ref, E)
    assert torch.equal(r.expert_offsets.cpu(), offs)
    n = int(offs[-1])
    assert n <= r.max_rows
    assert torch.equal(r.perm.cpu()[:n], perm)
    # weights: fp32, a few ulp (device expf vs host expf); exactly 1.0 where one choice survives
    assert torch.allclose(r.weight.cpu(), ref["weight"], atol=1e-6, rtol=0)
    one = (ref["slot"] >= 0).sum(1) == 1
    if one.any():
        assert torch.equal(r.weight.cpu()[one].max(1)[0], torch.ones(int(one.sum())))
    assert abs(float(r.l_aux) - float(ref["l_aux"])) <= 1e-5
    return r, ref


def test_routing_known_answer():
    logits = torch.tensor([[3., 2., 1.], [3., 1., 2.], [2., 3., 1.], [3., 2., 1.]])
    r, _ = _check_routing(logits, 2)
    assert r.exp_counts.tolist() == [4, 3, 1]
    assert r.slot.tolist() == [[0, 1], [1, 0], [0, -1], [-1, -1]]
    assert abs(float(r.l_aux) - 1.4936405) < 1e-5
    w = r.weight.cpu()
    assert float(w[2, 0]) == 1.0 and float(w[2, 1]) == 0.0 and float(w[3].abs().sum()) == 0.0
    assert abs(float(w[0, 0]) - 0.731059) < 2e-6 and abs(float(w[0, 1]) - 0.268941) < 2e-6


@pytest.mark.parametrize("S,E,cf,seed", [
    (1, 4, 2.0, 0), (2, 2, 1.0, 1), (63, 4, 1.5, 2), (64, 4, 1.5, 3), (255, 4, 1.0, 4), (256, 8, 1.5, 5),
    (257, 4, 0.5, 6), (704, 4, 1.5, 7), (2432, 4, 1.5, 8), (5000, 3, 0.75, 9), (22528, 4, 1.5, 10),
    (4099, 16, 1.25, 11), (3000, 64, 1.0, 12), (131072, 8, 1.0, 13)])
@pytest.mark.parametrize("noisy", [False, True])
def test_routing_bit_exact(S, E, cf, seed, noisy):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(S, E, generator=g)
    if seed % 2 == 1 and S > 8:                # capacity-drop stress: expert 0 overflows
        logits[: S // 2, 0] += 2.0
    noise = lit.gumbel_noise((S, E), 99) if noisy else None
    C = lit.capacity_of(S, E, cf * 2, 0)
    _check_routing(logits, C, noise)


def test_routing_no_drop_and_min_capacity():
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(777, 4, generator=g)
    logits[:, 1] += 1.5
    _check_routing(logits, 0, None, drop=False)
    _check_routing(logits, lit.capacity_of(777, 4, 0.1, 300), None)       # min_capacity wins


def test_routing_ties_and_extremes():
    S, E = 300, 4
    _check_routing(torch.zeros(S, E), lit.capacity_of(S, E, 3.0, 0))      # all equal: experts 0 then 1
    lg = torch.zeros(S, E)
    lg[:, 2] = 5.0
    lg[:, 1] = 1.0
    _check_routing(lg, lit.capacity_of(S, E, 3.0, 0))                      # everything to experts 2 and 1
    lg = torch.randn(S, E, generator=torch.Generator().manual_seed(1)).round()   # many exact ties
    _check_routing(lg, lit.capacity_of(S, E, 2.0, 0))
    lg = torch.randn(S, E, generator=torch.Generator().manual_seed(2)) * 60.0    # gates underflow to 0
    _check_routing(lg, lit.capacity_of(S, E, 3.0, 0))


def test_routing_fuzz_vs_index_oracle():
    """120 seeded random configurations against the index oracle, every field bit-exact: token counts around the kernels'
    own boundaries (64 / 1024 / 4096 tokens: one-launch decode form, single-block form, three-kernel form), odd expert
    counts, capacity factors from starving to no-drop, min_capacity, quantised logits (exact ties), skewed logits
    (overflowing experts), injected Gumbel noise, drop_tokens on and off."""
    import random
    rnd = random.Random(20240611)
    edges = [1, 2, 3, 63, 64, 65, 127, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097]
    for case in range(120):
        S = rnd.choice(edges) if rnd.random() < 0.5 else rnd.randint(1, 6000)
        E = rnd.choice([2, 3, 4, 4, 4, 5, 7, 8, 8, 12, 16, 17, 33, 64])
        g = torch.Generator().manual_seed(1000 + case)
        logits = torch.randn(S, E, generator=g) * rnd.choice([0.1, 1.0, 1.0, 4.0, 30.0])
        kind = rnd.random()
        if kind < 0.25:
            logits = (logits * 2).round() / 2                     # exact ties between experts
        elif kind < 0.5:
            logits[: max(1, S // 2), rnd.randrange(E)] += 2.5     # one expert overflows its capacity
        noise = lit.gumbel_noise((S, E), 7 + case) if rnd.random() < 0.5 else None
        drop = rnd.random() < 0.8
        cf = rnd.choice([0.1, 0.5, 1.0, 1.25, 1.5, 2.0, 4.0])
        C = lit.capacity_of(S, E, cf * 2, rnd.choice([0, 0, 4, 17])) if drop else 0
        try:
            _check_routing(logits, C, noise, drop)
        except AssertionError as e:
            raise AssertionError(f"case {case}: S={S} E={E} cf={cf} C={C} drop={drop} noisy={noise is not None} kind={kind:.2f}: {e}")


def test_capacity_host_formula_matches_oracle():
    from moe_llava_amd.sharded_moe import capacity_of
    for S in (1, 3, 5, 7, 704, 705, 2432, 22528, 54848, 131072):
        for E in (2, 3, 4, 8, 64):
            for cf in (0.25, 0.5, 1.0, 1.1, 1.25, 1.5, 2.0, 3.3):
                for mc in (0, 4, 16):
                    assert capacity_of(S, E, cf * 2, mc) == lit.capacity_of(S, E, cf * 2, mc), (S, E, cf, mc)


@pytest.mark.parametrize("wg_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("S,M,E", [(1, 64, 4), (37, 2048, 4), (704, 2048, 4), (513, 2560, 8), (100, 512, 16), (50, 256, 64)])
def test_gate_logits_numerics(S, M, E, wg_dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(S + M + E)
    x = torch.randn(S, M, generator=g).to(torch.bfloat16)
    wg = (torch.randn(E, M, generator=g) * 0.02).to(wg_dtype)
    got = ops.gate_logits(x.to(DEV), wg.to(DEV)).cpu()
    ref = (x.double() @ wg.double().t())
    tol = 1e-5 * float(ref.abs().max()) + 2e-6 * (M ** 0.5)
    assert float((got.double() - ref).abs().max()) <= tol
    # and against the fp32 torch op the reference uses
    ref32 = torch.nn.functional.linear(x.float(), wg.float())
    assert torch.allclose(got, ref32, atol=tol, rtol=0)


@pytest.mark.parametrize("dtype,wg_dtype", [(torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16),
                                            (torch.float16, torch.float32), (torch.float16, torch.float16)])
@pytest.mark.parametrize("S,M", [(4097, 2048), (22528, 2048), (13712, 2560), (4099, 2560), (9001, 2048)])
def test_gate_logits_streaming_form_is_bit_identical_and_accurate(S, M, dtype, wg_dtype):
    """Large S at M = 2048 / 2560 with 4 experts runs `gate_logits_stream_kernel` (persistent blocks, gate weight in LDS):
    the same logits bit for bit as the general kernel (MOE_NO_GATE_STREAM=1), which the routing parity relies on, and
    within the GEMV tolerance of fp64; odd S covers the clamped last row and waves without rows.  (fp16 activations with
    an fp16 gate weight stay on the general kernel, whose FMAs the compiler pairs into v_dot2c_f32_f16.)"""
    ops = _ops()
    E = 4
    g = torch.Generator().manual_seed(S + M)
    x = torch.randn(S, M, generator=g).to(dtype).to(DEV)
    wg = (torch.randn(E, M, generator=g) * 0.02).to(wg_dtype).to(DEV)
    got = ops.gate_logits(x, wg)
    with knob("MOE_NO_GATE_STREAM", "1"):
        general = ops.gate_logits(x, wg)
    assert torch.equal(got, general)
    ref = x.cpu().double() @ wg.cpu().double().t()
    tol = 1e-5 * float(ref.abs().max()) + 2e-6 * (M ** 0.5)
    assert float((got.cpu().double() - ref).abs().max()) <= tol


def _mk_experts(kind, M, I, E, seed, scale=0.05, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    mk = {"stablelm": lambda: StableLMMLPRef(M, I), "phi": lambda: PhiMLPRef(M, I),
          "qwen": lambda: QWenMLPRef(M, I), "qwen_bias": lambda: QWenMLPRef(M, I, bias=True)}[kind]
    exps = []
    for _ in range(E):
        e = mk().to(dtype)
        for p in e.parameters():
            p.data = (torch.randn(p.shape, generator=g) * scale).to(dtype)
        exps.append(e)
    return exps


def _build_pair(kind, M, I, E, S_shape, cf, seed, noisy=False, training=True, dtype=torch.bfloat16, use_residual=False,
                drop_tokens=True):
    """(oracle MoERef on CPU, product MoE on GPU) with identical parameters."""
    import copy
    import moe_llava_amd
    exps = _mk_experts(kind, M, I, E, seed, dtype=dtype)
    ref = lit.MoERef(M, exps[0], num_experts=E, k=2, capacity_factor=cf, eval_capacity_factor=2.0,
                     min_capacity=0, top2_2nd_expert_sampling=noisy, use_residual=use_residual, drop_tokens=drop_tokens)
    if use_residual:
        ref.coefficient.to(dtype)
    for dst, src in zip(ref.deepspeed_moe.experts.deepspeed_experts, exps):
        dst.load_state_dict(src.state_dict())
    g = torch.Generator().manual_seed(seed + 100)
    ref.deepspeed_moe.gate.wg.weight.data = torch.randn(E, M, generator=g) * 0.1
    # (residual mode keeps the prototype itself as `.mlp`, upstream's "same object, not a copy": each side gets its own)
    mine = moe_llava_amd.MoE(M, copy.deepcopy(exps[0]) if use_residual else exps[0], num_experts=E, k=2, capacity_factor=cf,
                             eval_capacity_factor=2.0, min_capacity=0, top2_2nd_expert_sampling=noisy, use_residual=use_residual,
                             drop_tokens=drop_tokens)
    missing = mine.load_state_dict(ref.state_dict(), strict=True)      # checkpoint-key ABI
    assert not missing.missing_keys and not missing.unexpected_keys
    for m in mine.deepspeed_moe.experts.deepspeed_experts:
        m.to(dtype)
    if use_residual:
        mine.coefficient.to(dtype)
        mine.mlp.to(dtype)
    mine = mine.to(DEV)
    ref.train(training)
    mine.train(training)
    x = torch.randn(*S_shape, M, generator=g).to(dtype)
    return ref, mine, x


@pytest.mark.parametrize("kind,M,I,E,shape,cf", [
    ("stablelm", 64, 128, 4, (2, 48), 1.0),
    ("stablelm", 128, 320, 4, (3, 111), 1.5),
    ("qwen", 128, 192, 4, (1, 300), 0.5),
    ("qwen_bias", 64, 64, 2, (1, 70), 1.5),
    ("phi", 128, 256, 4, (2, 130), 1.5),
    ("phi", 64, 576, 8, (1, 257), 1.0),
])
@pytest.mark.parametrize("noisy", [False, True])
def test_moe_layer_vs_literal_oracle(kind, M, I, E, shape, cf, noisy):
    """Whole block against the dense-mask oracle, given the kernel's own fp32 logits (the parity contract)."""
    ref, mine, x = _build_pair(kind, M, I, E, shape, cf, seed=7, noisy=noisy)
    S = x.numel() // M
    noise = lit.gumbel_noise((S, E), 99) if noisy else None
    captured = {}
    h = mine.deepspeed_moe.gate.wg.register_forward_hook(lambda m, i, o: captured.__setitem__("lg", o.detach().cpu()))
    if noisy:
        mine.deepspeed_moe.gate.noise_override = noise.to(DEV)
        ref.deepspeed_moe.gate.noise = noise
    with torch.no_grad():
        out, l_aux, exp_counts = mine(x.to(DEV))
    torch.cuda.synchronize()
    h.remove()
    assert isinstance(out, torch.Tensor) and out.shape == x.shape and out.dtype == torch.bfloat16
    assert l_aux.dim() == 0 and l_aux.dtype == torch.float32
    assert exp_counts.dtype == torch.int64 and exp_counts.shape == (E,)
    lg = captured["lg"]
    assert lg.shape == (S, E) and lg.dtype == torch.float32                # wg hook sees [S,E] fp32 logits
    ref_lg = torch.nn.functional.linear(x.reshape(-1, M).float(), ref.deepspeed_moe.gate.wg.weight.float())
    assert torch.allclose(lg, ref_lg, atol=1e-4, rtol=1e-4)
    ref.deepspeed_moe.gate.logits_override = lg
    with torch.no_grad():
        r_out, r_laux, r_counts = ref(x)
    assert torch.equal(exp_counts.cpu(), r_counts)
    assert abs(float(l_aux) - float(r_laux)) <= 1e-5
    assert_out_close(f"{kind} M{M} I{I} E{E} S{S} noisy={noisy}", out.cpu(), r_out)
    # rows of tokens that lost both choices are exactly zero
    slot = mine.deepspeed_moe.last_routing.slot.cpu()
    dead = (slot < 0).all(1)
    if dead.any():
        assert float(out.cpu().reshape(-1, M)[dead].float().abs().max()) == 0.0


def test_moe_layer_fuzz_vs_literal_oracle():
    """24 seeded random blocks (all expert kinds, token counts from a decode step to several row tiles, starving to
    generous capacity, with and without the Gumbel second choice) through the same check as above."""
    import random
    rnd = random.Random(99)
    for case in range(24):
        kind = rnd.choice(["stablelm", "phi", "qwen", "qwen_bias"])
        M, I, E = rnd.choice([64, 128, 256]), rnd.choice([64, 192, 320]), rnd.choice([2, 4, 4, 8])
        B = rnd.choice([1, 2, 3])
        T = rnd.choice([1, 2, 7, 31, 33, 64, 65, 129, 257, 300])
        cf = rnd.choice([0.25, 0.5, 1.0, 1.5, 2.0])
        noisy = rnd.random() < 0.5
        try:
            test_moe_layer_vs_literal_oracle(kind, M, I, E, (B, T), cf, noisy)
        except AssertionError as e:
            raise AssertionError(f"case {case}: {kind} M{M} I{I} E{E} shape {(B, T)} cf {cf} noisy {noisy}: {e}")


@pytest.mark.parametrize("kind,M,I,E,shape,cf,wg16", [
    ("stablelm", 128, 320, 4, (3, 111), 1.5, False),
    ("stablelm", 256, 384, 4, (1, 5), 1.0, True),          # decode size: the weight-streaming kernel
    ("qwen_bias", 64, 192, 8, (1, 300), 0.5, True),
    ("phi", 128, 256, 4, (2, 130), 1.5, True),
    ("phi", 64, 576, 8, (1, 257), 1.0, False),
])
@pytest.mark.parametrize("noisy", [False, True])
def test_moe_layer_fp16_vs_literal_oracle(kind, M, I, E, shape, cf, wg16, noisy):
    """The reference's inference loader builds the model in fp16 (moellava/model/builder.py:70, :475 model.to(float16);
    serve/cli.py:67-69): the same block in fp16 -- fp16 MFMA, fp16 rounding points, gate in fp32 from (optionally fp16)
    gate weights -- against the literal oracle run in fp16 on the CPU, to the same bars as bf16."""
    ref, mine, x = _build_pair(kind, M, I, E, shape, cf, seed=9, noisy=noisy, training=False, dtype=torch.float16)
    if wg16:                                               # model.to(torch.float16) casts the gate weight too
        mine = mine.to(torch.float16)
        ref.deepspeed_moe.gate.wg.weight.data = ref.deepspeed_moe.gate.wg.weight.data.to(torch.float16).float()
        assert mine.deepspeed_moe.gate.wg.weight.dtype == torch.float16
    S = x.numel() // M
    if noisy:
        noise = lit.gumbel_noise((S, E), 98)
        mine.deepspeed_moe.gate.noise_override = noise.to(DEV)
        ref.deepspeed_moe.gate.noise = noise
    with torch.no_grad():
        out, l_aux, exp_counts = mine(x.to(DEV))
    torch.cuda.synchronize()
    assert out.dtype == torch.float16 and out.shape == x.shape and l_aux.dtype == torch.float32
    lg = mine.deepspeed_moe.gate.last_logits.cpu()
    ref_lg = torch.nn.functional.linear(x.reshape(-1, M).float(), ref.deepspeed_moe.gate.wg.weight.float())
    assert lg.dtype == torch.float32 and torch.allclose(lg, ref_lg, atol=1e-4, rtol=1e-4)
    ref.deepspeed_moe.gate.logits_override = lg
    with torch.no_grad():
        r_out, r_laux, r_counts = ref(x)
    assert torch.equal(exp_counts.cpu(), r_counts) and abs(float(l_aux) - float(r_laux)) <= 1e-5
    r = mine.deepspeed_moe.last_routing
    want = ix.route_top2(lg, r.capacity, noise if noisy else None)
    assert torch.equal(r.slot.cpu(), want["slot"]) and torch.equal(r.expert_idx.cpu(), want["expert_idx"])
    # fp16 carries 3 more mantissa bits than bf16: the same bar is loose here, the printed numbers show the margin
    assert_out_close(f"fp16 {kind} M{M} I{I} E{E} S{S} noisy={noisy}", out.cpu(), r_out)


def test_fp16_is_forward_only_and_dtypes_must_agree():
    ref, mine, x = _build_pair("stablelm", 64, 128, 4, (1, 40), 1.0, seed=2, dtype=torch.float16)
    with pytest.raises(NotImplementedError):                  # training runs in bf16, as the reference's scripts do
        mine(x.to(DEV).requires_grad_(True))
    with pytest.raises(TypeError):                            # bf16 activations into fp16 experts
        with torch.no_grad():
            mine(x.to(DEV).to(torch.bfloat16))
    with pytest.raises(TypeError):
        with torch.no_grad():
            mine(x.to(DEV).float())


def test_top2_ignores_used_token_and_eval_ignores_jitter():
    """Upstream top2gating takes no used_token (TopKGate.forward drops it for k = 2; RePaMoE.forward passes *args
    through, llava_stablelm_moe.py:709), and a noisy-gate policy only acts while training."""
    ref, mine, x = _build_pair("stablelm", 64, 128, 4, (2, 40), 1.0, seed=5, training=False)
    used = torch.zeros(80, dtype=torch.long, device=DEV)
    with torch.no_grad():
        a, la, ca = mine(x.to(DEV))
        b, lb, cb = mine(x.to(DEV), used)
        mine.deepspeed_moe.gate.noisy_gate_policy = "Jitter"
        c, lc, cc = mine(x.to(DEV))                           # eval: the policy is ignored
    assert torch.equal(a, b) and torch.equal(ca, cb) and float(la) == float(lb)
    assert torch.equal(a, c) and torch.equal(ca, cc)
    assert mine.deepspeed_moe.gate.last_jitter_seed == 0


@pytest.mark.gpu
@pytest.mark.parametrize("S,M,E", [(37, 64, 4), (300, 2048, 4), (129, 512, 16)])
def test_gate_jitter_factors_and_logits(S, M, E):
    """noisy_gate_policy='Jitter': the factors are U(1 - 1e-2, 1 + 1e-2) samples that depend on (seed, token, column)
    only, and the gate kernel's logits are F.linear(x.float() * factors, wg) (upstream multiplicative_jitter)."""
    ops = _ops()
    g = torch.Generator().manual_seed(S + M)
    x = torch.randn(S, M, generator=g).to(torch.bfloat16)
    wg = torch.randn(E, M, generator=g) * 0.1
    seed = 0x1234_5678_9abc_def1
    J = ops.gate_jitter(seed, S, M, DEV).cpu()
    assert float(J.min()) >= 1.0 - 1e-2 and float(J.max()) <= 1.0 + 1e-2
    assert abs(float(J.mean()) - 1.0) < 2e-3 and float(J.std()) > 4e-3          # uniform on a 2e-2 interval: std 5.8e-3
    assert torch.equal(J[: S // 2], ops.gate_jitter(seed, S // 2, M, DEV).cpu())  # no dependence on the launch shape
    assert not torch.equal(J, ops.gate_jitter(seed + 1, S, M, DEV).cpu())
    lg = ops.gate_logits(x.to(DEV), wg.to(DEV), jitter_seed=seed).cpu()
    ref = (x.double() * J.double()) @ wg.double().T
    plain = ops.gate_logits(x.to(DEV), wg.to(DEV)).cpu()
    assert float((lg.double() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((plain.double() - ref).abs().max()) > 10 * float((lg.double() - ref).abs().max())


def test_wg_hooks_of_every_kind_fire_with_torch_semantics():
    """The reference captures gating logits with plain forward hooks on every nn.Linear named wg (moellava/utils.py:38-46).
    With any hook registered the gate kernel runs as wg's forward inside nn.Module.__call__: pre-hooks (which may replace
    the input), hooks registered with_kwargs, and torch's global module hooks all fire, and all see the fp32 [S,E] logits."""
    import torch.nn.modules.module as tm
    ref, mine, x = _build_pair("stablelm", 64, 128, 4, (2, 50), 1.0, seed=4)
    wg = mine.deepspeed_moe.gate.wg
    S = 100
    seen = {}
    with torch.no_grad():
        base, _, _ = mine(x.to(DEV))                            # no hooks: the one-call path
        base_lg = mine.deepspeed_moe.gate.last_logits.clone()
        hs = [wg.register_forward_pre_hook(lambda m, inp: seen.__setitem__("pre", tuple(inp[0].shape))),
              wg.register_forward_hook(lambda m, a, kw, o: seen.__setitem__("kw", (kw, o.dtype, tuple(o.shape))), with_kwargs=True),
              wg.register_forward_hook(lambda m, i, o: seen.__setitem__("plain", o.detach().cpu().clone())),
              tm.register_module_forward_hook(lambda m, i, o: seen.__setitem__("global", True) if m is wg else None)]
        try:
            out, _, _ = mine(x.to(DEV))
        finally:
            for h in hs:
                h.remove()
        assert seen["pre"] == (S, 64) and seen["kw"] == ({}, torch.float32, (S, 4)) and seen["global"] is True
        assert torch.equal(seen["plain"], base_lg.cpu()) and torch.equal(out, base)
        # a pre-hook that replaces the input changes the logits the routing uses (as it would for an eager nn.Linear)
        h = wg.register_forward_pre_hook(lambda m, inp: (inp[0] * 0,))
        try:
            mine(x.to(DEV))
        finally:
            h.remove()
        assert float(mine.deepspeed_moe.gate.last_logits.abs().max()) == 0.0
        # a hook that returns a replacement of the wrong kind is refused, not silently used
        h = wg.register_forward_hook(lambda m, i, o: o.to(torch.bfloat16))
        try:
            with pytest.raises(TypeError):
                mine(x.to(DEV))
        finally:
            h.remove()
        assert "forward" not in wg.__dict__                     # the instance-level forward never leaks
        again, _, _ = mine(x.to(DEV))
        assert torch.equal(again, base)


def test_eval_mode_uses_eval_capacity_and_s1_decode():
    ref, mine, x = _build_pair("stablelm", 64, 128, 4, (1, 1), 1.0, seed=3, training=False)
    with torch.no_grad():
        out, l_aux, cnt = mine(x.to(DEV))
    lg = _ops().gate_logits(x.reshape(-1, 64).to(DEV), mine.deepspeed_moe.gate.wg.weight).cpu()
    ref.deepspeed_moe.gate.logits_override = lg
    with torch.no_grad():
        r_out, r_laux, r_cnt = ref(x)
    assert mine.deepspeed_moe.last_routing.capacity == lit.capacity_of(1, 4, 4.0, 0) == 1
    assert torch.equal(cnt.cpu(), r_cnt)
    assert_out_close("decode S=1", out.cpu(), r_out)


@pytest.mark.parametrize("kind,M,I", [("stablelm", 2048, 5632), ("phi", 2560, 10240)])
def test_cfg1_block_vs_index_oracle(kind, M, I):
    """BASELINE config 1 shape (1 x (576 img + 128 text) tokens, 4 experts, top-2, cf 1.5) against the index-form
    oracle on a sample of tokens (the CPU MLP over every row takes too long for the CPU suite budget)."""
    E, S = 4, 704
    ref, mine, x = _build_pair(kind, M, I, E, (1, S), 1.5, seed=11)
    for e in ref.deepspeed_moe.experts.deepspeed_experts:          # keep activations O(1)
        for p in e.parameters():
            p.data = (p.data.float() * 0.4).to(torch.bfloat16)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(DEV)
    with torch.no_grad():
        out, l_aux, cnt = mine(x.to(DEV))
    torch.cuda.synchronize()
    lg = _ops().gate_logits(x.reshape(-1, M).to(DEV), mine.deepspeed_moe.gate.wg.weight).cpu()
    C = lit.capacity_of(S, E, 3.0, 0)
    assert C == 528
    r = ix.route_top2(lg, C)
    assert torch.equal(cnt.cpu(), r["exp_counts"])
    assert torch.equal(mine.deepspeed_moe.last_routing.slot.cpu(), r["slot"])
    rt = mine.deepspeed_moe.last_routing
    # random tokens + the tokens at the experts' first / last rows and around the 128-row tile edges of this launch
    sample = torch.unique(torch.cat([torch.randperm(S, generator=torch.Generator().manual_seed(0))[:64],
                                     edge_tokens(rt.expert_offsets.cpu(), rt.perm, 2, 128, seed=1)]))
    sub = {k: (v[sample] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == S else v) for k, v in r.items()}
    with torch.no_grad():
        want = ix.moe_forward_index(x.reshape(-1, M)[sample], sub, list(ref.deepspeed_moe.experts.deepspeed_experts))
    got = out.cpu().reshape(-1, M)[sample]
    assert_out_close(f"cfg1 {kind}", got, want)


def _full_size_check(kind, M, I, E, B, T, cf, stress=False, dtype=torch.bfloat16):
    """Size-independent properties at a BASELINE shape instead of a full CPU run: determinism, counts, capacity,
    perm = inverse of (expert, slot), routing equal to the index oracle given the kernel's logits, rows that lost
    both choices exactly zero, and a sample of output rows against the oracle expert MLPs."""
    import moe_llava_amd
    from tests.expert_mlps import PhiMLP, QWenMLP, StableLMMLP
    mk = {"stablelm": (StableLMMLP, StableLMMLPRef), "phi": (PhiMLP, PhiMLPRef), "qwen": (QWenMLP, QWenMLPRef)}[kind]
    torch.manual_seed(0)
    moe = moe_llava_amd.MoE(M, mk[0](M, I), num_experts=E, k=2, capacity_factor=cf,
                            eval_capacity_factor=2.0, min_capacity=0, top2_2nd_expert_sampling=False)
    moe = moe.to(dtype)
    for p in moe.deepspeed_moe.experts.parameters():
        p.data.normal_(0, 0.02)
    moe.deepspeed_moe.gate.wg.weight.data.normal_(0, 0.2)
    if stress:                                         # SURVEY 8d capacity-drop stress: expert 0 overflows
        moe.deepspeed_moe.gate.wg.weight.data[0] += 0.05
    moe = moe.to(DEV).train()
    x = torch.randn(B, T, M, device=DEV).to(dtype)
    x[:, : (T * 4) // 5] += 0.5                        # "image" rows skew the router (drops happen)
    with torch.no_grad():
        out1, l1, c1 = moe(x)
        r = moe.deepspeed_moe.last_routing
        out2, l2, c2 = moe(x)
    torch.cuda.synchronize()
    S = B * T
    assert torch.equal(out1, out2) and torch.equal(c1, c2) and float(l1) == float(l2)        # deterministic
    assert int(c1.sum()) == 2 * S
    C = lit.capacity_of(S, E, cf * 2, 0)
    assert r.capacity == C
    kept = r.kept_counts.cpu()
    assert int(kept.max()) <= C
    offs = r.expert_offsets.cpu()
    n = int(offs[-1])
    perm_raw = r.perm.cpu()
    perm = perm_raw[:n].long()
    assert perm.unique().numel() == n                                   # each (token, choice) at most once
    slot, idx = r.slot.cpu().long(), r.expert_idx.cpu().long()
    tok, ch = perm // 2, perm % 2
    pos = torch.arange(n)
    assert torch.equal(offs[idx[tok, ch]].long() + slot[tok, ch], pos)  # perm is the inverse of (expert, slot)
    assert int((slot >= 0).sum()) == n
    # routing equals the index-form oracle given the kernel's logits
    lg = _ops().gate_logits(x.reshape(-1, M), moe.deepspeed_moe.gate.wg.weight).cpu()
    ref = ix.route_top2(lg, C)
    assert torch.equal(slot.int(), ref["slot"]) and torch.equal(idx.int(), ref["expert_idx"])
    assert abs(float(l1) - float(ref["l_aux"])) <= 1e-5
    dead = (slot < 0).all(1)
    o = out1.reshape(-1, M)
    if dead.any():
        assert float(o[dead.to(DEV)].float().abs().max()) == 0.0
    if stress:
        assert int((slot < 0).sum()) > 0                                # the stress really drops
    # sampled rows against the oracle MLPs: random tokens, plus the tokens whose compact rows are the first / last row of
    # an expert, lie on either side of a row-tile boundary (256-row tiles when an expert has > 512 rows, else 128) or in
    # an expert's last partial tile -- where a tile-edge or clamp bug at the real K would show
    bm = 256 if (n + E - 1) // E > 512 else 128
    sample = torch.unique(torch.cat([torch.randperm(S, generator=torch.Generator().manual_seed(1))[:32],
                                     edge_tokens(offs, perm_raw, 2, bm, seed=2, per_expert=4)]))
    assert sample.numel() >= 40
    sub = {k: (v[sample] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == S else v) for k, v in ref.items()}
    cpu_experts = [mk[1](M, I).to(dtype) for _ in range(E)]
    for ce, ge in zip(cpu_experts, moe.deepspeed_moe.experts.deepspeed_experts):
        ce.load_state_dict({k: v.cpu() for k, v in ge.state_dict().items()})
    with torch.no_grad():
        want = ix.moe_forward_index(x.reshape(-1, M)[sample.to(DEV)].cpu(), sub, cpu_experts)
    got = o[sample.to(DEV)].cpu()
    assert_out_close(f"{kind} M{M} I{I} E{E} S{S} {dtype} sampled rows", got, want)
    return r


def test_full_size_properties_cfg2():
    """BASELINE config 2: StableLM-1.6B-4e block, batch 32 -> S = 22 528."""
    r = _full_size_check("stablelm", 2048, 5632, 4, 32, 704, 1.5)
    assert r.capacity == 16896


def test_full_size_properties_cfg2_fp16():
    """Config 2's shape in the reference's inference dtype (fp16, builder.py:70): the 256 x 256 fp16 MFMA tile."""
    _full_size_check("stablelm", 2048, 5632, 4, 32, 704, 1.5, dtype=torch.float16)


def test_full_size_properties_cfg3_phi_rank_share():
    """BASELINE config 3 (Phi2-384: M 2560, I 10 240, gelu_new + biases): one GPU's share at EP = 4, 16 x 857 rows."""
    _full_size_check("phi", 2560, 10240, 4, 16, 857, 1.5)


def test_full_size_properties_cfg4_qwen():
    """BASELINE config 4 (Qwen-1.8B-4e: I = 5504, not a multiple of 256; 4 x 576 + 128 tokens per sample)."""
    _full_size_check("qwen", 2048, 5504, 4, 4, 2432, 1.5)


def test_full_size_properties_cfg5_drop_stress():
    """BASELINE config 5 at its REAL per-GPU size: 1 M tokens / 8 GPUs = 131 072 tokens, 8 experts of Phi-2 width,
    capacity-drop stress (cf 1.0, expert 0 overloaded): 262 144 compact rows at most, 27 TFLOP."""
    r = _full_size_check("phi", 2560, 10240, 8, 128, 1024, 1.0, stress=True)
    assert r.S == 131072 and r.capacity == 32768


def test_cfg3_phi_stack_32_layers():
    """BASELINE config 3 read literally: a 32-layer stack of Phi2-width MoE blocks (M 2560, I 10 240, gelu_new + biases,
    4 experts, top-2, cf 1.5) over one GPU's share of batch 64 x (729 + 128) tokens at EP = 4: S = 13 712.  Layers are
    chained the way the Phi decoder layer does it (llava_phi_moe.py:121-140: x + moe(layernorm(x)), attention left out).
    Checked: the whole stack is deterministic; EVERY layer's routing equals the index oracle given that layer's
    kernel logits (bit-exact), l_aux within 1e-5; sampled output rows of four layers against the oracle expert MLPs."""
    import moe_llava_amd
    from tests.expert_mlps import PhiMLP
    M, I, E, L, B, T = 2560, 10240, 4, 32, 16, 857
    S = B * T
    assert S == 13712
    torch.manual_seed(3)
    layers = []
    for l in range(L):
        moe = moe_llava_amd.MoE(M, PhiMLP(M, I), num_experts=E, k=2, capacity_factor=1.5, eval_capacity_factor=2.0,
                                min_capacity=0, top2_2nd_expert_sampling=False).to(torch.bfloat16)
        for p in moe.deepspeed_moe.experts.parameters():
            p.data.normal_(0, 0.02)
        moe.deepspeed_moe.gate.wg.weight.data.normal_(0, 0.2)
        layers.append(moe.to(DEV).train())
    x0 = torch.randn(B, T, M, device=DEV).to(torch.bfloat16)
    x0[:, :729] += 0.5                                  # SigLIP image rows skew the router
    C = lit.capacity_of(S, E, 3.0, 0)
    assert C == 10284

    def run(keep):
        x = x0
        with torch.no_grad():
            for l, moe in enumerate(layers):
                h = torch.nn.functional.layer_norm(x.float(), (M,)).to(torch.bfloat16)
                out, l_aux, cnt = moe(h)
                if keep is not None:
                    r = moe.deepspeed_moe.last_routing
                    keep.append((h, out, float(l_aux), cnt.clone(), r.slot.clone(), r.expert_idx.clone(),
                                 r.expert_offsets.cpu().clone(), r.perm.cpu().clone()))
                x = x + out
        return x

    kept = []
    y1 = run(kept)
    y2 = run(None)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and bool(torch.isfinite(y1.float()).all())
    sampled = {0, 10, 21, L - 1}
    smp0 = torch.randperm(S, generator=torch.Generator().manual_seed(2))[:24]
    dropped = 0
    for l, (h, out, l_aux, cnt, slot, idx, offs_l, perm_l) in enumerate(kept):
        moe = layers[l]
        lg = _ops().gate_logits(h.reshape(-1, M), moe.deepspeed_moe.gate.wg.weight).cpu()
        ref = ix.route_top2(lg, C)
        assert torch.equal(slot.cpu(), ref["slot"]) and torch.equal(idx.cpu(), ref["expert_idx"]), f"layer {l}"
        assert torch.equal(cnt.cpu(), ref["exp_counts"]) and abs(l_aux - float(ref["l_aux"])) <= 1e-5, f"layer {l}"
        dropped += int((ref["slot"] < 0).sum())
        if l in sampled:
            smp = torch.unique(torch.cat([smp0, edge_tokens(offs_l, perm_l, 2, 256, seed=l, per_expert=2)]))
            sub = {k: (v[smp] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == S else v) for k, v in ref.items()}
            cpu_experts = [PhiMLPRef(M, I).to(torch.bfloat16) for _ in range(E)]
            for ce, ge in zip(cpu_experts, moe.deepspeed_moe.experts.deepspeed_experts):
                ce.load_state_dict({k: v.cpu() for k, v in ge.state_dict().items()})
            with torch.no_grad():
                want = ix.moe_forward_index(h.reshape(-1, M)[smp.to(DEV)].cpu(), sub, cpu_experts)
            assert_out_close(f"cfg3 stack layer {l} sampled rows", out.reshape(-1, M)[smp.to(DEV)].cpu(), want)
    print(f"[cfg3 stack] {L} layers x S={S}: dropped choices over the stack = {dropped}")


def test_experts_forward_reference_layout():
    """Experts.forward([ep, E_local, C, M]) (upstream layout) equals the per-expert oracle MLPs."""
    import moe_llava_amd
    M, I, El, C = 64, 128, 3, 37
    exps = _mk_experts("stablelm", M, I, El, seed=5)
    ex = moe_llava_amd.Experts(exps[0], El, "ep_size_1")
    for d, s in zip(ex.deepspeed_experts, exps):
        d.load_state_dict(s.state_dict())
    ex = ex.to(DEV)
    x = torch.randn(2, El, C, M, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16)
    y = ex(x.to(DEV)).cpu()
    with torch.no_grad():
        want = torch.cat([exps[i](x[:, i:i + 1]) for i in range(El)], dim=1)
    assert_out_close("Experts.forward", y, want)
    for p in ex.parameters():
        assert p.allreduce is False and p.group_name == "ep_size_1"


def test_state_dict_views_survive_load_and_inplace_update():
    """Expert params are views of the packed buffers: load_state_dict / in-place updates are seen by the kernel."""
    ref, mine, x = _build_pair("stablelm", 64, 128, 4, (1, 64), 2.0, seed=21)
    with torch.no_grad():
        o1, _, _ = mine(x.to(DEV))
        sd = {k: v.clone() for k, v in mine.state_dict().items()}
        for k in sd:
            if "down_proj" in k:
                sd[k] = sd[k] * 2
        mine.load_state_dict(sd)
        o2, _, _ = mine(x.to(DEV))
        for e in mine.deepspeed_moe.experts.deepspeed_experts:
            e.down_proj.weight.mul_(0.5)
        o3, _, _ = mine(x.to(DEV))
    assert torch.allclose(o2.float(), 2 * o1.float(), rtol=2e-2, atol=1e-3)
    assert torch.equal(o3, o1)


def test_errors_are_loud():
    import moe_llava_amd
    from moe_llava_amd._lib import MoEKernelError
    from tests.expert_mlps import StableLMMLP
    ops = _ops()
    with pytest.raises(MoEKernelError):
        ops.gate_logits(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64))       # CPU tensor
    with pytest.raises(TypeError):
        ops.gate_logits(torch.zeros(4, 64, device=DEV), torch.zeros(4, 64, device=DEV))     # fp32 activations
    with pytest.raises(MoEKernelError):
        ops.gate_logits(torch.zeros(4, 60, dtype=torch.bfloat16, device=DEV), torch.zeros(4, 60, device=DEV))  # M%8
    moe = moe_llava_amd.MoE(64, StableLMMLP(64, 128), num_experts=4, k=2).to(DEV)           # fp32 experts
    with pytest.raises(TypeError):
        moe(torch.zeros(1, 8, 64, dtype=torch.bfloat16, device=DEV))
    with pytest.raises(NotImplementedError):
        moe_llava_amd.MoE(64, torch.nn.Linear(64, 64), num_experts=2, k=2).to(torch.bfloat16).to(DEV)(
            torch.zeros(1, 8, 64, dtype=torch.bfloat16, device=DEV))


def _philox4x32_10(c0, c1, k0, k1):
    """numpy Philox4x32-10 (test-side restatement of the kernel's counter-based generator)."""
    import numpy as np
    c = [np.asarray(c0, np.uint64), np.asarray(c1, np.uint64), np.zeros_like(c0, np.uint64), np.zeros_like(c0, np.uint64)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [(hi1 ^ c[1] ^ k0) & mask, lo1, (hi0 ^ c[3] ^ k1) & mask, lo0]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c


@pytest.mark.parametrize("S,E", [(704, 4), (3000, 8), (22528, 4), (9000, 16)])
def test_in_kernel_gumbel_noise(S, E):
    """noise_seed mode: Gumbel noise generated inside the routing kernel (Philox keyed by the seed)."""
    import numpy as np
    ops = _ops()
    g = torch.Generator().manual_seed(S + E)
    logits = torch.randn(S, E, generator=g)
    C = lit.capacity_of(S, E, 3.0, 0)
    seed = 0x1234_5678_9ABC_DEF1
    dl = logits.to(DEV)
    a = ops.routing_top2(dl, C, None, True, noise_seed=seed)
    b = ops.routing_top2(dl, C, None, True, noise_seed=seed)
    c = ops.routing_top2(dl, C, None, True, noise_seed=seed + 1)
    d = ops.routing_top2(dl, C, None, True)                                   # deterministic branch
    torch.cuda.synchronize()
    assert torch.equal(a.expert_idx, b.expert_idx) and torch.equal(a.slot, b.slot) and torch.equal(a.weight, b.weight)
    assert not torch.equal(a.expert_idx, c.expert_idx)                        # another seed, another sample
    assert torch.equal(a.expert_idx[:, 0], d.expert_idx[:, 0])                # first choice never sees the noise
    assert int((a.expert_idx[:, 1] != d.expert_idx[:, 1]).sum()) > S // 20    # the second choice does
    assert int(a.exp_counts.sum()) == 2 * S and (a.expert_idx[:, 0] != a.expert_idx[:, 1]).all()
    # replicate the generator on the host and hand the same noise to the oracle
    s_idx = np.repeat(np.arange(S, dtype=np.uint64), (E + 3) // 4)
    grp = np.tile(np.arange((E + 3) // 4, dtype=np.uint64), S)
    words = _philox4x32_10(s_idx, grp, seed & 0xFFFFFFFF, seed >> 32)
    bits = np.stack(words, 1).reshape(S, -1)[:, :E].astype(np.uint32)
    u = ((bits >> 9).astype(np.float32) + np.float32(0.5)) * np.float32(1.1920928955078125e-07)
    noise = torch.from_numpy(-np.log(-np.log(u)).astype(np.float32))
    assert abs(float(noise.mean()) - 0.5772) < 0.05 and abs(float(noise.var()) - 1.6449) < 0.15   # Gumbel(0,1) moments
    ref = ix.route_top2(logits.clone(), C, noise)
    same = (a.expert_idx.cpu() == ref["expert_idx"]).all(1)
    # logf differs by an ulp between host and device: allow a handful of near-tie tokens, nothing systematic
    assert int((~same).sum()) <= max(2, S // 2000), int((~same).sum())
    if bool(same.all()):
        assert torch.equal(a.slot.cpu(), ref["slot"]) and torch.equal(a.exp_counts.cpu(), ref["exp_counts"])


def test_fused_and_three_kernel_routing_agree():
    """S <= 4096 takes the single-launch kernel, larger S the three-kernel path: same semantics on a prefix."""
    ops = _ops()
    E = 8
    g = torch.Generator().manual_seed(77)
    big = torch.randn(6000, E, generator=g)
    big[:3000, 0] += 2.0
    for S in (1, 64, 1023, 1024, 1025, 4096):
        C = lit.capacity_of(S, E, 2.0, 0)
        _check_routing(big[:S].contiguous(), C)
    _check_routing(big[:4097].contiguous(), lit.capacity_of(4097, E, 2.0, 0))


@pytest.mark.parametrize("kind,M,I,E", [("stablelm", 256, 384, 4), ("phi", 128, 512, 4), ("qwen_bias", 64, 192, 8)])
@pytest.mark.parametrize("S", [1, 2, 5, 17, 32])
def test_decode_sizes_skinny_kernel_vs_oracle_and_tiled(kind, M, I, E, S):
    """Decode steps (<= 64 compact rows) run the weight-streaming GEMV-style kernel: outputs match the oracle to the
    usual bar and the tiled kernel to bf16 rounding (the K split changes the fp32 summation order)."""
    import os
    ref, mine, x = _build_pair(kind, M, I, E, (S, 1), 1.0, seed=31, training=False)
    with torch.no_grad():
        out, l_aux, cnt = mine(x.to(DEV))
        with knob("MOE_NO_SKINNY", "1"):
            out_tiled, _, _ = mine(x.to(DEV))
        # the fc2 launch picks 1 or 2 column groups per block from the launch size; both give the same bits
        forced = []
        for nw in ("1", "2"):
            with knob("MOE_SKINNY_NW", nw):
                forced.append(mine(x.to(DEV))[0])
    torch.cuda.synchronize()
    assert torch.equal(forced[0], forced[1]) and torch.equal(forced[0], out)
    assert mine.deepspeed_moe.last_routing.max_rows <= 64
    ref.deepspeed_moe.gate.logits_override = mine.deepspeed_moe.gate.last_logits.cpu()
    with torch.no_grad():
        r_out, r_laux, r_cnt = ref(x)
    assert torch.equal(cnt.cpu(), r_cnt)
    err, scale = assert_out_close(f"decode {kind} S{S}", out.cpu(), r_out)
    # two kernels of ours (K split differently): each within the bar of the oracle, so within twice the bar of each other
    assert_out_close(f"decode {kind} S{S} skinny vs tiled", out.cpu(), out_tiled.cpu(), atol_of_scale=8e-3, rtol=2e-2)


@pytest.mark.parametrize("S,E,wg_bf16", [(1, 4, False), (5, 4, True), (33, 8, False), (64, 4, False), (64, 8, True)])
def test_decode_gate_and_routing_in_one_launch_is_bit_identical(S, E, wg_bf16):
    """For S <= 64 and E <= 8 the one-call block runs the gate GEMV and the routing as one launch: logits, routing and
    output must equal the separate launches bit for bit (MOE_NO_GATE_ROUTE_FUSION=1 selects those)."""
    import os
    ops = _ops()
    from moe_llava_amd import capacity_of
    M, I = 256, 192
    g = torch.Generator().manual_seed(S * 10 + E)
    x = torch.randn(S, M, generator=g).to(torch.bfloat16).to(DEV)
    wg = (torch.randn(E, M, generator=g) * 0.2).to(torch.bfloat16 if wg_bf16 else torch.float32).to(DEV)
    w1 = (torch.randn(E, 2 * I, M, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    w2 = (torch.randn(E, M, I, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    cap = capacity_of(S, E, 2.0, 0)
    seed = 4242 + S
    out, lg, r = ops.block_top2(x, wg, cap, None, seed, True, 0, w1, None, w2, None)
    with knob("MOE_NO_GATE_ROUTE_FUSION", "1"):
        out2, lg2, r2 = ops.block_top2(x, wg, cap, None, seed, True, 0, w1, None, w2, None)
    torch.cuda.synchronize()
    assert torch.equal(lg, ops.gate_logits(x, wg)) and torch.equal(lg, lg2)
    for name in ("expert_idx", "slot", "weight", "exp_counts", "kept_counts", "expert_offsets", "l_aux"):
        assert torch.equal(getattr(r, name), getattr(r2, name)), name
    n = int(r.expert_offsets[-1])
    assert torch.equal(r.perm[:n], r2.perm[:n]) and torch.equal(out, out2)


@pytest.mark.parametrize("epilogue", [1, 0, 2])
def test_persistent_launch_equals_one_block_per_tile(epilogue):
    """Launches of >= 4096 big tiles run as 256 persistent blocks that walk the tile list (for SwiGLU with the next
    tile's first fills issued during the epilogue); MOE_NO_PERSIST=1 keeps one block per tile.  Same bits, for the
    SwiGLU (1), plain (0) and gelu_new + bias (2) epilogues."""
    import os
    from moe_llava_amd import _lib
    lib = _lib.load()
    M, I, E, R = 1024, 5632, 4, 25000                      # 98+ row tiles x 44 column tiles > 4096 tiles
    g = torch.Generator().manual_seed(3)
    x = torch.randn(R // 2, M, generator=g).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(E, 2 * I, M, generator=g) * 0.03).to(torch.bfloat16).to(DEV)
    b1 = (torch.randn(E, 2 * I, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    cnt = torch.tensor([9000, 1, 7000, R - 16001])
    offs = torch.zeros(E + 1, dtype=torch.int32)
    offs[1:] = torch.cumsum(cnt, 0).int()
    offs = offs.to(DEV)
    perm = torch.randperm(R, generator=g).int().to(DEV)    # token * 2 + choice, any order
    st = torch.cuda.current_stream().cuda_stream
    N = I if epilogue == 1 else 2 * I                      # output columns (SwiGLU halves the 2I weight rows)
    outs = []
    for env in (None, "1"):
        with knob("MOE_NO_PERSIST", env):
            h = torch.zeros(R, N, dtype=torch.bfloat16, device=DEV)
            rc = lib.moe_grouped_gemm_fwd(epilogue, x.data_ptr(), M, perm.data_ptr(), 1, w1.data_ptr(),
                                          b1.data_ptr() if epilogue == 2 else None, offs.data_ptr(), E, E,
                                          R, M, N, -1, h.data_ptr(), N, None, 0, 0, st)
            assert rc == 0, lib.moe_last_error()
            torch.cuda.synchronize()
            outs.append(h)
    assert torch.equal(outs[0], outs[1])
    # spot check against torch on a few rows of every group
    for e in range(E):
        lo = int(offs[e])
        rows = torch.arange(lo, min(lo + 3, int(offs[e + 1])), device=DEV)
        if rows.numel() == 0:
            continue
        xr = x[(perm[rows] >> 1).long()].float()
        ab = xr @ w1[e].float().t()
        if epilogue == 1:
            ab = ab.to(torch.bfloat16).float()
            want = (torch.nn.functional.silu(ab[:, :I]).to(torch.bfloat16).float() * ab[:, I:]).to(torch.bfloat16)
        elif epilogue == 0:
            want = ab.to(torch.bfloat16)
        else:
            want = torch.nn.functional.gelu((ab + b1[e].float()).to(torch.bfloat16).float(), approximate="tanh").to(torch.bfloat16)
        assert torch.allclose(outs[0][rows].float(), want.float(), rtol=3e-2, atol=3e-2)


def _chunk_major_pos(offs, C):
    """Host restatement of the chunk-major order of the pipelined expert-parallel exchange: chunk c
    holds rows [c*k/C, (c+1)*k/C) of every expert's k kept rows, experts in order inside a chunk."""
    E = len(offs) - 1
    pos = [0] * offs[-1]
    nxt = 0
    for c in range(C):
        for e in range(E):
            k = offs[e + 1] - offs[e]
            for i in range(k * c // C, k * (c + 1) // C):
                pos[offs[e] + i] = nxt
                nxt += 1
    return torch.tensor(pos, dtype=torch.long)


@pytest.mark.gpu
@pytest.mark.parametrize("counts", [[5, 0, 3, 9], [1, 1, 1, 1], [0, 0, 0, 0], [700, 33, 0, 1290, 2, 64, 0, 511]])
@pytest.mark.parametrize("C", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_ep_chunk_rows_order_is_bit_exact(counts, C, dtype):
    """moe_ep_chunk_rows_fwd through the C ABI: send order (gathered through perm), unpack, and their round trip -- pure data
    movement, so bit-exact; experts with no rows and experts with fewer rows than chunks included."""
    ops = _ops()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(sum(counts) * 31 + C)
    E, n, M = len(counts), sum(counts), 192
    offs = [0]
    for k in counts:
        offs.append(offs[-1] + k)
    S = max(n, 1)
    x = torch.randn(S, M, generator=g).to(dtype)
    tok = torch.randperm(S, generator=g)[:n]
    perm = (2 * tok + torch.randint(0, 2, (n,), generator=g)).to(torch.int32)   # 2 * token + choice, as the routing kernel writes it
    max_rows = n + 7                                                            # the buffers are sized for the worst case
    perm_dev = torch.cat([perm, torch.full((7,), -1, dtype=torch.int32)]).to(dev)
    offs_dev = torch.tensor(offs, dtype=torch.int32, device=dev)
    pos = _chunk_major_pos(offs, C)
    send = ops.ep_chunk_rows(x.to(dev), perm_dev, 1, offs_dev, C, False, max_rows)
    torch.cuda.synchronize()
    want = torch.empty(n, M, dtype=dtype)
    want[pos] = x[tok]
    assert torch.equal(send[:n].cpu(), want)
    # no gather index: rows already compact
    rows = torch.randn(max(max_rows, 1), M, generator=g).to(dtype)
    send2 = ops.ep_chunk_rows(rows.to(dev), None, 0, offs_dev, C, False, max_rows)
    want2 = torch.empty(n, M, dtype=dtype)
    want2[pos] = rows[:n]
    assert torch.equal(send2[:n].cpu(), want2)
    # unpack is the inverse map, into a caller-provided buffer
    out = torch.full((max(max_rows, 1), M), 7.0, dtype=dtype, device=dev)
    back = ops.ep_chunk_rows(send2, None, 0, offs_dev, C, True, max_rows, out=out)
    torch.cuda.synchronize()
    assert back.data_ptr() == out.data_ptr()
    assert torch.equal(back[:n].cpu(), rows[:n])
    assert torch.all(back[n:].cpu() == 7.0)                                     # rows past the kept count are not touched
