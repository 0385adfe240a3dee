"""HIP-graph replay of the block (moe_llava_amd.graph.GraphedMoE) against the eager one-call path.

The replayed step must (a) give exactly (bit for bit) what the eager path gives for the seed the device held before
the replay and (b) advance that seed on the device, so successive replays draw different second-expert noise.  The
eager path itself is held to the oracle in test_gpu_parity.py (in-kernel noise: test_in_kernel_gumbel_noise).  The
device-seed routing entry point is also checked on its own through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
MASK = 0xFFFFFFFFFFFFFFFF


def _next_seed(s):
    nx = (s * 6364136223846793005 + 1442695040888963407) & MASK
    return nx or 1


def _build(kind, M, I, E, S, training, seed=5):
    from tests.test_gpu_parity import _build_pair
    ref, mine, x = _build_pair(kind, M, I, E, (S, 1), 1.0, seed=seed, noisy=True, training=training)
    return ref, mine, x.to(DEV)


def test_dseed_routing_entry_point_matches_value_seed_and_advances():
    from moe_llava_amd import _lib, ops
    lib = _lib.load()
    S, E, C = 37, 4, 20
    torch.manual_seed(0)
    logits = torch.randn(S, E, device=DEV)
    seed = 0x1234567890ABCDEF
    want = ops.routing_top2(logits, C, None, True, noise_seed=seed)
    seed_dev = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed], dtype=torch.int64, device=DEV)
    ws_bytes = lib.moe_routing_workspace_bytes(S, E)
    r = ops.Routing.allocate(S, E, 2, min(2 * S, E * C), C, ws_bytes, logits.device)
    st = torch.cuda.current_stream().cuda_stream
    for step in range(2):
        rc = lib.moe_routing_top2_fwd_dseed(logits.data_ptr(), seed_dev.data_ptr(), S, E, C, 1, r.ptr("expert_idx"),
                                            r.ptr("slot"), r.ptr("weight"), r.ptr("exp_counts"), r.ptr("kept_counts"),
                                            r.ptr("expert_offsets"), r.ptr("perm"), r.ptr("l_aux"), r.ptr("workspace"),
                                            ws_bytes, st)
        assert rc == 0, lib.moe_last_error()
        torch.cuda.synchronize()
        if step == 0:
            for name in ("expert_idx", "slot", "weight", "exp_counts", "kept_counts", "expert_offsets", "l_aux"):
                assert torch.equal(getattr(r, name), getattr(want, name)), name
            assert int(seed_dev.item()) & MASK == _next_seed(seed)
        else:
            want2 = ops.routing_top2(logits, C, None, True, noise_seed=_next_seed(seed))
            assert torch.equal(r.expert_idx, want2.expert_idx) and torch.equal(r.slot, want2.slot)
            assert int(seed_dev.item()) & MASK == _next_seed(_next_seed(seed))
    # sizes beyond the single-launch kernel are refused, nothing is launched
    big = torch.randn(5000, E, device=DEV)
    rb = ops.Routing.allocate(5000, E, 2, 10000, 5000, lib.moe_routing_workspace_bytes(5000, E), big.device)
    rc = lib.moe_routing_top2_fwd_dseed(big.data_ptr(), seed_dev.data_ptr(), 5000, E, 5000, 1, rb.ptr("expert_idx"),
                                        rb.ptr("slot"), rb.ptr("weight"), rb.ptr("exp_counts"), rb.ptr("kept_counts"),
                                        rb.ptr("expert_offsets"), rb.ptr("perm"), rb.ptr("l_aux"), rb.ptr("workspace"),
                                        lib.moe_routing_workspace_bytes(5000, E), st)
    assert rc == -2 and b"S <= 4096" in lib.moe_last_error()


@pytest.mark.parametrize("kind,M,I,E,S,training", [
    ("stablelm", 128, 320, 4, 1, False),
    ("stablelm", 128, 320, 4, 32, False),
    ("phi", 128, 256, 4, 7, False),
    ("qwen", 128, 192, 4, 48, True),
])
def test_graph_replay_equals_eager(kind, M, I, E, S, training):
    from moe_llava_amd import ops
    from moe_llava_amd.graph import GraphedMoE
    ref, mine, x = _build(kind, M, I, E, S, training)
    g = GraphedMoE(mine, x, seed=777)
    layer, gate = mine.deepspeed_moe, mine.deepspeed_moe.gate
    assert "noise_seed_dev" not in gate.__dict__          # the eager path is back to host seeds
    cf = gate.capacity_factor if training else gate.eval_capacity_factor
    cap = ops_capacity(S, E, cf)
    act, w1, b1, w2, b2 = layer.experts.packed(x.device)
    seen = []
    for step in range(3):
        xs = (x + 0.25 * step).to(torch.bfloat16)
        seed = int(g.seed_dev.item()) & MASK
        out, l_aux, cnt = g(xs)
        torch.cuda.synchronize()
        assert int(g.seed_dev.item()) & MASK == _next_seed(seed), "the replay must advance the device seed"
        e_out, e_lg, e_r = ops.block_top2(xs.reshape(-1, M), gate.wg.weight, cap, None, seed, gate.drop_tokens, act, w1,
                                         b1, w2, b2, layer.experts.n_gated)
        torch.cuda.synchronize()
        assert torch.equal(out.reshape(-1, M), e_out), f"step {step}: replay differs from the eager call"
        assert torch.equal(g.routing.expert_idx, e_r.expert_idx) and torch.equal(g.routing.slot, e_r.slot)
        assert torch.equal(cnt, e_r.exp_counts) and torch.equal(l_aux, e_r.l_aux)
        seen.append(seed)
    assert len(set(seen)) == 3
    with pytest.raises(ValueError):
        g(torch.zeros(S + 1, 1, M, dtype=torch.bfloat16, device=DEV))


def ops_capacity(S, E, cf):
    from moe_llava_amd import capacity_of
    return capacity_of(S, E, cf * 2, 0)


def test_graph_refuses_what_it_cannot_capture():
    from moe_llava_amd.graph import GraphedMoE
    _, mine, x = _build("stablelm", 64, 128, 4, 4, False)
    mine.deepspeed_moe.gate.noisy_gate_policy = "RSample"
    with pytest.raises(NotImplementedError):
        GraphedMoE(mine, x)
    mine.deepspeed_moe.gate.noisy_gate_policy = None
    with pytest.raises(TypeError):
        GraphedMoE(mine, x.float())
