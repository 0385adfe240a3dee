"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header declares, the
drop-in module keeps the reference's constructor / attribute tree / checkpoint keys, host-side capacity and
group arithmetic, weight packing as views, and loud failure without a GPU."""
import os
import re
import sys

import pytest
import torch

import moe_llava_amd
from moe_llava_amd import _lib, groups
from tests.expert_mlps import PhiMLP, QWenMLP, StableLMMLP
from moe_llava_amd.sharded_moe import capacity_of
from oracle import ds_moe_literal as lit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "moe_gfx950.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(moe_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from moe_llava_amd.build import build_library
        build_library()
    lib = _lib.load()
    declared = _header_symbols()
    assert declared, "no declarations parsed from the header"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.moe_abi_version() == _lib.ABI_VERSION == 8
    assert lib.moe_splitk_workspace_bytes() == 64 << 20
    assert lib.moe_routing_workspace_bytes(22528, 4) > 0


def test_argument_validation_happens_before_any_launch():
    """Negative return codes = nothing launched; safe to call without a GPU."""
    lib = _lib.load()
    assert lib.moe_gate_logits_fwd(None, None, 0, None, 4, 64, 4, 0, 0, None) == -1
    assert b"null" in lib.moe_last_error()
    assert lib.moe_gate_logits_fwd(16, 16, 0, 16, 4, 60, 4, 0, 0, None) == -2       # M % 8
    assert lib.moe_gate_logits_fwd(8, 16, 0, 16, 4, 64, 4, 1, 0, None) == -3        # alignment
    assert lib.moe_gate_logits_fwd(16, 16, 1, 16, 4, 64, 4, 1, 0, None) == -1       # bf16 gate weight with fp16 activations
    assert lib.moe_gate_logits_fwd(16, 16, 0, 16, 4, 64, 4, 2, 0, None) == -1       # unknown dtype
    assert lib.moe_routing_top2_fwd(16, None, 0, 4, 1, 2, 1, 16, 16, 16, 16, 16, 16, 16, 16, 16, 1 << 20, None) == -2  # E<2
    assert lib.moe_routing_top2_fwd(16, None, 0, 4, 4, 2, 1, 16, 16, 16, 16, 16, 16, 16, 16, 16, 8, None) == -4  # workspace
    assert lib.moe_expert_ffn_fwd(7, 16, 64, None, 0, 16, None, 16, None, 16, 4, 4, 8, 64, 64, -1, 16, 16, None, 0, 0, None) == -1
    assert lib.moe_expert_ffn_fwd(0, 16, 64, None, 0, 16, None, 16, None, 16, 4, 4, 8, 64, 100, -1, 16, 16, None, 0, 1, None) == -2
    assert lib.moe_combine_fwd(16, 16, 16, 16, 16, 4, 64, 3, 16, 0, None) == -2
    # weight-gradient kernel: 32-bit byte offsets inside a group -> more than 4 GiB of rows is refused, not mis-addressed
    assert lib.moe_expert_ffn_bwd(0, 16, 2048, None, 0, 16, 16, 16, 16, 16, 16, 4, 4, 200000, 2048, 5632, -1, 16, 16, None,
                                  16, None, 16, 1 << 40, None) == -2
    assert b"4 GiB" in lib.moe_last_error()


def test_capacity_matches_oracle_formula():
    for S in (1, 2, 5, 7, 9, 704, 705, 857, 2432, 22528, 54848, 131072, 1048576):
        for E in (2, 3, 4, 5, 8, 64):
            for cf in (0.1, 0.25, 0.5, 0.75, 1.0, 1.1, 1.2, 1.25, 1.5, 2.0, 3.3):
                for mc in (0, 4, 16):
                    assert capacity_of(S, E, cf * 2, mc) == lit.capacity_of(S, E, cf * 2, mc), (S, E, cf, mc)
    assert capacity_of(704, 4, 3.0, 0) == 528 and capacity_of(22528, 4, 3.0, 0) == 16896


def test_constructor_attribute_tree_and_checkpoint_keys():
    # the nine kwargs the reference passes (llava_stablelm_moe.py:550-560)
    moe = moe_llava_amd.MoE(2048 // 32, expert=StableLMMLP(64, 128), num_experts=4, ep_size=1, k=2,
                            capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0, use_residual=False)
    assert moe.expert_group_name == "ep_size_1" and moe.num_experts == 4 and moe.num_local_experts == 4
    assert isinstance(moe.deepspeed_moe.experts.deepspeed_experts, torch.nn.ModuleList)
    assert isinstance(moe.deepspeed_moe.gate.wg, torch.nn.Linear) and moe.deepspeed_moe.gate.wg.bias is None
    assert moe.deepspeed_moe.gate.wg.weight.shape == (4, 64)
    assert moe.deepspeed_moe.gate.top2_2nd_expert_sampling is True          # upstream default, never overridden
    keys = set(moe.state_dict().keys())
    want = {"deepspeed_moe.gate.wg.weight"} | {
        f"deepspeed_moe.experts.deepspeed_experts.{i}.{n}.weight" for i in range(4)
        for n in ("gate_proj", "up_proj", "down_proj")}
    assert keys == want
    for p in moe.deepspeed_moe.experts.parameters():
        assert p.allreduce is False and p.group_name == "ep_size_1"
    assert not hasattr(moe.deepspeed_moe.gate.wg.weight, "allreduce")
    # names with 'wg' are what moellava/utils.py:38-46 hooks
    assert any("wg" in n and isinstance(m, torch.nn.Linear) for n, m in moe.named_modules())
    # same keys as the oracle's restatement of upstream
    ref = lit.MoERef(64, StableLMMLP(64, 128), num_experts=4, k=2)
    assert set(ref.state_dict().keys()) == keys
    # experts start as copies of the dense MLP (weight-copy assert, llava_stablelm_moe.py:561-564)
    proto = PhiMLP(64, 128)
    moe2 = moe_llava_amd.MoE(64, proto, num_experts=2, k=2, use_residual=True)
    for e in moe2.deepspeed_moe.experts.deepspeed_experts:
        for k, v in e.state_dict().items():
            assert torch.allclose(v, proto.state_dict()[k])
    assert moe2.mlp is proto and moe2.coefficient.weight.shape == (2, 64)
    with pytest.raises(AssertionError):
        moe_llava_amd.MoE(64, proto, num_experts=3, ep_size=2)


def test_expert_adapters_and_packing_views_on_cpu():
    from moe_llava_amd import ops
    from moe_llava_amd.experts import Experts, describe_expert
    s = describe_expert(StableLMMLP(64, 128))
    assert s.act == ops.ACT_SWIGLU and [l.weight.shape for l in s.fc1] == [(128, 64), (128, 64)]
    q = QWenMLP(64, 128)
    s = describe_expert(q)
    assert s.act == ops.ACT_SWIGLU and s.fc1[0] is q.w2 and s.fc1[1] is q.w1 and s.fc2 is q.c_proj
    s = describe_expert(PhiMLP(64, 128))
    assert s.act == ops.ACT_GELU_NEW and s.fc1[0].bias is not None
    with pytest.raises(NotImplementedError):
        describe_expert(torch.nn.Linear(4, 4))
    ex = Experts(PhiMLP(64, 128).to(torch.bfloat16), 3, "ep_size_1")
    act, w1, b1, w2, b2 = ex.packed(torch.device("cpu"))
    assert w1.shape == (3, 128, 64) and w2.shape == (3, 64, 128) and b1.shape == (3, 128) and b2.shape == (3, 64)
    e1 = ex.deepspeed_experts[1]
    assert e1.fc1.weight.data_ptr() == w1[1].data_ptr() and e1.fc2.bias.data_ptr() == b2[1].data_ptr()
    with torch.no_grad():
        e1.fc1.weight.fill_(3.0)
    assert float(w1[1].float().mean()) == 3.0
    assert ex.packed(torch.device("cpu"))[1] is w1                      # no re-pack while pointers are stable
    ex.to(torch.bfloat16)
    sd = ex.state_dict()
    assert "deepspeed_experts.2.fc2.bias" in sd
    with pytest.raises(TypeError):
        Experts(PhiMLP(64, 128), 1, "g").packed(torch.device("cpu"))   # fp32 experts are refused


def test_expert_parallel_rank_layout():
    ep, edp = groups.expert_parallel_ranks(8, 4)
    assert ep == [[0, 1, 2, 3], [4, 5, 6, 7]] and edp == [[0, 4], [1, 5], [2, 6], [3, 7]]
    ep, edp = groups.expert_parallel_ranks(8, 8)
    assert ep == [list(range(8))] and edp == [[i] for i in range(8)]
    with pytest.raises(ValueError):
        groups.expert_parallel_ranks(6, 4)


def test_shim_installs_under_deepspeed_import_path():
    from moe_llava_amd import shim
    saved = {k: v for k, v in sys.modules.items() if k == "deepspeed" or k.startswith("deepspeed.")}
    try:
        shim.install()
        from deepspeed.moe.layer import MoE as ShimMoE          # the import line of llava_stablelm_moe.py:29
        assert ShimMoE is moe_llava_amd.MoE
        from deepspeed.moe.sharded_moe import TopKGate as G
        assert G is moe_llava_amd.TopKGate
    finally:
        for k in [k for k in sys.modules if k == "deepspeed" or k.startswith("deepspeed.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_no_gpu_means_loud_failure_not_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    moe = moe_llava_amd.MoE(64, StableLMMLP(64, 128), num_experts=4, k=2).to(torch.bfloat16)
    with pytest.raises(_lib.MoEKernelError):
        moe(torch.zeros(1, 8, 64, dtype=torch.bfloat16))


@pytest.mark.parametrize("shape", [(0, 7, 64), (3, 0, 64)])
def test_call_without_tokens_matches_the_literal_oracle(shape):
    """No tokens: nothing to launch, so this runs without a GPU.  Same result as upstream's gating on empty logits
    (oracle/ds_moe_literal.py): empty output, zero counts, NaN `l_aux` (a mean over zero tokens)."""
    mine = moe_llava_amd.MoE(64, StableLMMLP(64, 128), num_experts=4, k=2, capacity_factor=1.5).to(torch.bfloat16)
    ref = lit.MoERef(64, StableLMMLP(64, 128), num_experts=4, k=2, capacity_factor=1.5).to(torch.bfloat16)
    ref.deepspeed_moe.gate.noise = torch.zeros(0, 4)   # the Gumbel noise of zero tokens
    x = torch.zeros(shape, dtype=torch.bfloat16)
    for train in (False, True):
        mine.train(train); ref.train(train)
        xm = x.clone().requires_grad_(train)
        out, l_aux, cnt = mine(xm)
        r_out, r_laux, r_cnt = ref(x.clone())
        assert out.shape == r_out.shape == x.shape and out.dtype == r_out.dtype
        assert torch.isnan(l_aux) and torch.isnan(r_laux) and l_aux.dtype == r_laux.dtype
        assert cnt.dtype == r_cnt.dtype and cnt.tolist() == r_cnt.tolist() == [0, 0, 0, 0]
        if train:
            out.sum().backward()
            assert xm.grad.shape == x.shape


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "moe_llava_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_repa_expert_adapter_padding_and_reparam_on_cpu():
    """RePaMLP (llava_stablelm_moe.py:613-655): prefix mask -> n_gated; reparam() shrinks the projections in place of
    the same storage address, which must still trigger a re-pack; widths that are not multiples of 64 are padded."""
    from tests.expert_mlps import RePaMLP
    from moe_llava_amd.experts import Experts, describe_expert
    from oracle.expert_mlps import RePaMLPRef
    for names in ("stablelm", "qwen"):
        m = RePaMLP(64, 200, names).to(torch.bfloat16)
        assert describe_expert(m).n_gated == -1                       # ratio 1.0 = ordinary SwiGLU
        m.adjust_gated_ratio(0.5)
        assert describe_expert(m).n_gated == 100
        ex = Experts(m, 2, "ep_size_1")
        act, w1, b1, w2, b2 = ex.packed(torch.device("cpu"))
        assert w1.shape == (2, 512, 64) and w2.shape == (2, 64, 256) and ex.n_gated == 100
        assert float(w1[:, 200:256].abs().max()) == 0 and float(w2[:, :, 200:].abs().max()) == 0
        r = RePaMLPRef(64, 200, names).to(torch.bfloat16)
        r.adjust_gated_ratio(0.5)
        r.load_state_dict(ex.deepspeed_experts[0].state_dict())
        x = torch.randn(3, 5, 64).to(torch.bfloat16)
        assert torch.equal(ex.deepspeed_experts[0](x), r(x))          # product mirror == oracle restatement
        f1 = [e.reparam() for e in ex.deepspeed_experts]
        f2 = r.reparam()
        assert torch.equal(f1[0], f2) and f1[0].shape == (64, 64)
        act, w1b, _, w2b, _ = ex.packed(torch.device("cpu"))
        assert w1b is not w1 and w1b.shape == (2, 256, 64) and w2b.shape == (2, 64, 128) and ex.n_gated == -1
        assert torch.equal(ex.deepspeed_experts[0](x), r(x))
    bad = RePaMLP(64, 128).to(torch.bfloat16)
    bad.adjust_gated_ratio(0.5)
    bad.mask[100] = True
    with pytest.raises(NotImplementedError):
        Experts(bad, 1, "g").packed(torch.device("cpu"))


def test_packed_leaves_externally_owned_parameters_alone():
    """An engine that flattens parameters into one buffer and updates that buffer in place (DeepSpeed ZeRO-1/2 bf16,
    scripts/zero2.json) owns the memory: packed() must neither re-point the parameters away from it nor miss its
    updates -- before and after the first pack."""
    import warnings
    from moe_llava_amd.experts import Experts

    def flatten(ex):
        ps = list(ex.parameters())
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.bfloat16)
        off = 0
        for p in ps:
            flat[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        return flat, ps

    for flatten_first in (True, False):
        ex = Experts(StableLMMLP(64, 128).to(torch.bfloat16), 3, "ep_size_1")
        for p in ex.parameters():
            p.data.normal_(0, 0.1)
        if not flatten_first:
            ex.packed(torch.device("cpu"))                         # parameters become views of the packed buffers
        flat, ps = flatten(ex)
        ptrs = [p.data_ptr() for p in ps]
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            act, w1, b1, w2, b2 = ex.packed(torch.device("cpu"))
        assert any("owned by someone else" in str(x.message) for x in w)
        assert [p.data_ptr() for p in ps] == ptrs                  # still views of the engine's flat buffer
        e1 = ex.deepspeed_experts[1]
        assert torch.equal(w1[1, :128], e1.gate_proj.weight.data) and torch.equal(w2[1], e1.down_proj.weight.data)
        before = w1.float().mean().item()
        flat.add_(1.0)                                             # the engine's in-place optimizer step
        act, w1b, _, w2b, _ = ex.packed(torch.device("cpu"))
        assert w1b is w1 and abs(w1b.float().mean().item() - (before + 1.0)) < 2e-2
        assert torch.equal(w2b[2], ex.deepspeed_experts[2].down_proj.weight.data)
        assert [p.data_ptr() for p in ps] == ptrs
        # .to() / load_state_dict style replacement (parameters own their storage again): re-pointed as before
        for p in ps:
            p.data = p.data.clone()
        act, w1c, _, _, _ = ex.packed(torch.device("cpu"))
        assert w1c is not w1 and ex.deepspeed_experts[0].gate_proj.weight.data_ptr() == w1c[0].data_ptr()
        assert not ex._external
