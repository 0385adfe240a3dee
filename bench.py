#!/usr/bin/env python3
"""MoE-layer throughput bench (BASELINE.json metric: MoE-layer tokens/s on MI355X, top-2, 4 / 8 experts).

A "step" is one forward of the sparse-MoE FFN block (gate -> top-2 routing with capacity drop -> grouped expert
FFN -> weighted combine) over one synthetic batch that is already resident in HBM, on the module's DEFAULT path
(the one-call entry point moe_block_top2_fwd at N = 1, the expert-parallel exchange at N > 1).

  --config 2 (default): BASELINE config 2 -- MoE-LLaVA-StableLM-1.6B-4e block (d=2048, expert width 5632 SwiGLU,
          4 experts, top-2, capacity factor 1.5, training-mode capacity), batch 32 x (576 image + 128 text) tokens.
  --config 1 / 3 / 4 / 5: the other BASELINE configs as per-GPU shapes (see CONFIGS).
  N > 1 : the same block and per-GPU batch on every rank (weak scaling); the experts shard expert-parallel over
          min(N, E) consecutive ranks (x N/ep replicas), token dispatch / combine by RCCL all-to-all.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  `roofline`       the dominant kernel (fc1 grouped GEMM), timed with HIP events between the launches in a pass of its
                   own right after the headline loop (events inside the headline loop would force the four-call path);
  `cpu_baseline`   the oracle on the host cores: the index-form restatement on config 2 itself (value) and the literal
                   dense-mask restatement on one 704-token sample (literal_config1);
  `other_configs`  (N = 1, default config only) configs 1, 3, 4, 5 and decode steps S = 1 / 32, a few timed steps each,
                   small shapes with 12 / 5 layers' weights cycled so that they come from HBM as in the real model.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA peak, bf16 and fp16 alike (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0

# Per-GPU shapes of the BASELINE.json configs.  `layers`: MoE layers whose weights are cycled through (the small
# shapes are weight-streaming bound: one layer's 277 MB would otherwise sit in the 256 MiB Infinity Cache between calls,
# which a 12-MoE-layer model never sees).
CONFIGS = {
    1: dict(tag="config 1: StableLM-1.6B-4e MoE block, 1 x (576 img + 128 text) tokens", kind="stablelm", M=2048, I=5632,
            E=4, cf=1.5, B=1, n_img=576, n_txt=128, layers=12, stress=False),
    2: dict(tag="BASELINE config 2: StableLM-1.6B-4e MoE block forward, batch 32 x (576 img + 128 text) tokens",
            kind="stablelm", M=2048, I=5632, E=4, cf=1.5, B=32, n_img=576, n_txt=128, layers=1, stress=False),
    3: dict(tag="config 3: Phi2-2.7B-4e-384 MoE layer (gelu_new + bias), one GPU's share of batch 64 x (729 img + 128 "
                "text) tokens at EP=4 (16 samples)", kind="phi", M=2560, I=10240, E=4, cf=1.5, B=16, n_img=729, n_txt=128,
            layers=1, stress=False),
    4: dict(tag="config 4: Qwen-1.8B-4e MoE block, 1 x (4 x 576 img + 128 text) tokens", kind="qwen", M=2048, I=5504, E=4,
            cf=1.5, B=1, n_img=2304, n_txt=128, layers=5, stress=False),
    5: dict(tag="config 5: synthetic 8-expert Phi2-width block, one GPU's share of 1M tokens at EP=8 (131072 tokens), "
                "capacity factor 1.0, capacity-drop stress (+2.0 on expert 0's logit for half of the tokens)", kind="phi",
            M=2560, I=10240, E=8, cf=1.0, B=128, n_img=512, n_txt=512, layers=1, stress=True),
}
TOPK = 2


def _u_img(M):
    u = torch.randn(M, generator=torch.Generator().manual_seed(4321))
    return u / u.norm() * (M ** 0.5)


def make_inputs(cfg, batch: int, device, seed: int, dtype=torch.bfloat16):
    """text rows ~ N(0,1); image rows N(0,1) + 0.5*u_img (one fixed direction with |u|^2 = M, gives the
    modality-skewed routing)."""
    M, n_img, n_txt = cfg["M"], cfg["n_img"], cfg["n_txt"]
    u = _u_img(M)
    if batch * (n_img + n_txt) <= 64 * 704:
        g = torch.Generator(device="cpu").manual_seed(seed)
        x = torch.randn(batch, n_img + n_txt, M, generator=g)
        x[:, :n_img] += 0.5 * u
        return x.to(dtype).to(device)
    g = torch.Generator(device=device).manual_seed(seed)       # large shapes: generate on the device
    x = torch.randn(batch, n_img + n_txt, M, generator=g, device=device)
    x[:, :n_img] += 0.5 * u.to(device)
    return x.to(dtype)


def _proto(cfg):
    from tests.expert_mlps import PhiMLP, QWenMLP, StableLMMLP
    return {"stablelm": StableLMMLP, "phi": PhiMLP, "qwen": QWenMLP}[cfg["kind"]](cfg["M"], cfg["I"])


def build_layers(cfg, device, ep_size: int, ep_rank: int, seed: int = 1234, dtype=torch.bfloat16, training=True):
    """`layers` MoE blocks built with the reference's nine constructor kwargs (llava_stablelm_moe.py:550-560,
    finetune_moe.sh:3-13); random-init weights at HF's init scale 0.02.  Every rank draws the weights of all E experts
    in the same order and keeps its own shard, so an expert has the same weights whatever the EP degree."""
    import moe_llava_amd
    out = []
    for li in range(cfg["layers"]):
        moe = moe_llava_amd.MoE(cfg["M"], expert=_proto(cfg).to(dtype), num_experts=cfg["E"], ep_size=ep_size, k=TOPK,
                                capacity_factor=cfg["cf"], eval_capacity_factor=2.0, min_capacity=0, use_residual=False)
        g = torch.Generator().manual_seed(seed + 17 * li)
        wg = torch.randn(cfg["E"], cfg["M"], generator=g) * 0.02
        if cfg["stress"]:
            wg[0] += (4.0 / cfg["M"]) * _u_img(cfg["M"])      # image rows (x + 0.5 u): +2.0 on expert 0's logit
        moe.deepspeed_moe.gate.wg.weight.data = wg
        moe = moe.to(device)
        gd = torch.Generator(device=device).manual_seed(seed + 1000 * (li + 1))
        el = moe.num_local_experts
        mods = list(moe.deepspeed_moe.experts.deepspeed_experts)
        for e in range(cfg["E"]):
            le = e - ep_rank * el
            for p in mods[le if 0 <= le < el else 0].parameters():
                v = torch.randn(p.shape, generator=gd, device=device) * 0.02
                if 0 <= le < el:
                    p.data = v.to(dtype)
        out.append(moe.train(training))
    return out


def flops_per_row(cfg):
    return (6.0 if cfg["kind"] != "phi" else 4.0) * cfg["M"] * cfg["I"]


def weight_bytes(cfg, n_experts):
    nproj = 3 if cfg["kind"] != "phi" else 2
    b = n_experts * nproj * cfg["M"] * cfg["I"] * 2.0
    if cfg["kind"] == "phi":
        b += n_experts * (cfg["I"] + cfg["M"]) * 2.0
    return b


def time_steps(layers, x, steps, warmup, fence):
    """W untimed + K timed forward steps, cycling the layers; returns the seconds the K steps took."""
    L = len(layers)
    with torch.no_grad():
        for i in range(warmup):
            layers[i % L](x)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            layers[i % L](x)
        fence()
    return time.perf_counter() - t0


def summarize(cfg, S, kept_pairs, active_experts, seconds_per_step):
    """tokens/s, achieved TFLOP/s and TB/s (algorithmic flops / bytes of SURVEY 8d) and the fraction of the roofline
    that bounds the shape: max(flops / MFMA peak, bytes / HBM peak) / time."""
    flops = kept_pairs * flops_per_row(cfg) + 2.0 * S * cfg["M"] * cfg["E"]
    bytes_ = weight_bytes(cfg, active_experts) + 2.0 * S * cfg["M"] * 2.0 + S * cfg["E"] * 4.0
    t_mfma = flops / (PEAK_BF16_TFLOPS * 1e12)
    t_hbm = bytes_ / (PEAK_HBM_GBS * 1e9)
    return {"S": S, "ms_per_step": seconds_per_step * 1e3, "tokens_per_s": S / seconds_per_step,
            "tflops": flops / seconds_per_step / 1e12, "tb_per_s": bytes_ / seconds_per_step / 1e12,
            "bound": "mfma" if t_mfma >= t_hbm else "hbm", "roofline_floor_us": max(t_mfma, t_hbm) * 1e6,
            "frac_of_roofline": max(t_mfma, t_hbm) / seconds_per_step, "frac_of_mfma_peak": t_mfma / seconds_per_step,
            "frac_of_hbm_peak": t_hbm / seconds_per_step, "kept_pairs_per_token": kept_pairs / S,
            "algorithmic_gflop": flops / 1e9, "algorithmic_mbytes": bytes_ / 1e6}


def run_other_configs(device, fence):
    """Configs 1, 3, 4, 5 and decode steps on this GPU: a few timed steps each (rank 0, N = 1 only)."""
    res = {}
    for cid in (1, 3, 4, 5):
        cfg = CONFIGS[cid]
        layers = build_layers(cfg, device, 1, 0, seed=4000 + cid)
        x = make_inputs(cfg, cfg["B"], device, seed=77 + cid)
        S = x.shape[0] * x.shape[1]
        steps, warm = (10, 3) if cid in (3, 5) else (48, 12)
        sec = time_steps(layers, x, steps, warm, fence) / steps
        with torch.no_grad():
            layers[0](x)
        r = layers[0].deepspeed_moe.last_routing
        kept = float(r.expert_offsets[-1])
        d = summarize(cfg, S, kept, cfg["E"], sec)
        d.update(workload=cfg["tag"], layers_cycled=cfg["layers"], steps=steps, warmup=warm,
                 dropped_choices=int(2 * S - kept), exp_counts=layers[0].deepspeed_moe.exp_counts.tolist())
        res[f"config{cid}"] = d
        del layers, x, r
        torch.cuda.empty_cache()
    res.update(run_decode(device, fence, (1, 32)))
    return res


def run_decode(device, fence, sizes, steps=120):
    """Decode steps: StableLM block in eval mode (eval_capacity_factor 2.0), 12 layers' weights cycled."""
    res = {}
    cfg = CONFIGS[1]
    layers = build_layers(cfg, device, 1, 0, seed=5000, training=False)
    for S in sizes:
        x = make_inputs(cfg, 1, device, seed=9)[:, :S].reshape(S, 1, cfg["M"]).contiguous()
        sec = time_steps(layers, x, steps, 24, fence) / steps
        act = 0
        for m in layers:
            with torch.no_grad():
                m(x)
            act += int((m.deepspeed_moe.exp_counts > 0).sum())
        d = summarize(cfg, S, 2.0 * S, act / len(layers), sec)
        d.update(workload=f"decode step, StableLM-1.6B-4e block, S = {S} tokens, eval capacity (idle experts' weights are "
                          f"not part of the algorithmic bytes)", layers_cycled=cfg["layers"], steps=steps, warmup=24,
                 active_experts_mean=act / len(layers))
        res[f"decode_S{S}"] = d
    del layers
    torch.cuda.empty_cache()
    return res


def cpu_baseline(state, budget_s: float = 6.0, gpu_moe=None, device=None):
    """The oracle timed on the host cores (checker code as a reported baseline, never shipped): the index-form
    restatement on the headline's own workload (config 2, `value`), and the literal dense-mask restatement on one
    704-token sample (config 1: S=704, C=528; `literal_config1`) with its labelled S^2 extrapolation."""
    from oracle import ds_moe_literal as lit
    from oracle.expert_mlps import StableLMMLPRef
    cfg = CONFIGS[2]
    M, I, E, CF = cfg["M"], cfg["I"], cfg["E"], cfg["cf"]
    torch.manual_seed(0)
    ref = lit.MoERef(M, StableLMMLPRef(M, I).to(torch.bfloat16), num_experts=E, ep_size=1, k=TOPK,
                     capacity_factor=CF, eval_capacity_factor=2.0, min_capacity=0, use_residual=False)
    ref.load_state_dict(state, strict=True)
    ref.train()
    x = make_inputs(cfg, 1, "cpu", seed=1234)
    S = x.shape[0] * x.shape[1]
    ref.deepspeed_moe.gate.noise = lit.gumbel_noise((S, E), 99)    # upstream default: 2nd-expert sampling on
    times = []
    with torch.no_grad():
        ref(x)                                               # warm-up (thread pool, allocator)
        t_all = time.perf_counter()
        while time.perf_counter() - t_all < budget_s and len(times) < 2000:
            t0 = time.perf_counter()
            ref(x)
            times.append(time.perf_counter() - t0)
    t = statistics.median(times)
    parity = None
    if gpu_moe is not None:
        # parity gates beside the perf number: the HIP block on the same 704-token sample against the
        # oracle fed the kernel's fp32 logits and the same Gumbel noise
        gate = gpu_moe.deepspeed_moe.gate
        gate.noise_override = ref.deepspeed_moe.gate.noise.to(device)
        try:
            with torch.no_grad():
                g_out, g_laux, g_cnt = gpu_moe(x.to(device))
            torch.cuda.synchronize()
            lg = gate.last_logits.float().cpu()
            r = gpu_moe.deepspeed_moe.last_routing
            from oracle import ds_moe_index as ix
            want = ix.route_top2(lg.clone(), r.capacity, ref.deepspeed_moe.gate.noise, True)
            ref.deepspeed_moe.gate.logits_override = lg
            with torch.no_grad():
                r_out, r_laux, r_cnt = ref(x)
            ref.deepspeed_moe.gate.logits_override = None
            top2 = torch.topk(torch.softmax(lg, 1), 2, dim=1).values
            d = (g_out.cpu().float() - r_out.float()).abs()
            scale = float(r_out.float().abs().max())
            parity = {"sample": f"S={S}, C={r.capacity}, oracle fed the kernel's logits and the same noise",
                      "routing_bit_exact": bool(torch.equal(r.expert_idx.cpu(), want["expert_idx"])
                                                and torch.equal(r.slot.cpu(), want["slot"])
                                                and torch.equal(g_cnt.cpu(), r_cnt)
                                                and torch.equal(r.kept_counts.cpu(), want["kept_counts"])),
                      "l_aux_abs_err": abs(float(g_laux) - float(r_laux)),
                      "out_max_abs_err": float(d.max()), "out_ref_max_abs": scale,
                      "out_within_bar": bool((d <= 4e-3 * scale + 1e-2 * r_out.float().abs()).all()),
                      "out_bar": "|d| <= 4e-3 * max|ref| + 1e-2 * |ref| (tests/_parity.py)",
                      "tie_sensitive_tokens": int(((top2[:, 0] - top2[:, 1]) < 1e-5).sum())}
        finally:
            gate.noise_override = None
    # ---- the headline's own workload on the CPU: index-form restatement (the dense [S,E,C] masks of the literal form
    #      would take 6 GB at S = 22528), same block, same batch shape ----
    index_cfg2 = None
    try:
        from oracle import ds_moe_index as ix
        import torch.nn.functional as F
        xb = make_inputs(cfg, cfg["B"], "cpu", seed=1234).reshape(-1, M)
        Sb = xb.shape[0]
        Cb = lit.capacity_of(Sb, E, CF * 2, 0)
        wgc = ref.deepspeed_moe.gate.wg.weight.detach().float()
        experts = list(ref.deepspeed_moe.experts.deepspeed_experts)
        noise_b = lit.gumbel_noise((Sb, E), 99)
        t_idx = []
        with torch.no_grad():
            for it in range(4):
                t0 = time.perf_counter()
                route = ix.route_top2(F.linear(xb.float(), wgc), Cb, noise_b, True)
                ix.moe_forward_index(xb, route, experts)
                dt = time.perf_counter() - t0
                if it > 0 or dt > 8.0:                       # first pass = warm-up unless a pass is already long
                    t_idx.append(dt)
                if sum(t_idx) > 12.0:
                    break
        ti = statistics.median(t_idx)
        index_cfg2 = {"tokens_per_s": Sb / ti, "seconds": ti, "runs": len(t_idx), "S": Sb, "capacity": Cb,
                      "kept_pairs": int(route["kept_counts"].sum())}
    except MemoryError:
        pass
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    literal = {"tokens_per_s": S / t, "seconds": t, "runs": len(times), "S": S,
               "sample": f"literal dense-mask oracle (restated deepspeed MoE: [S,E,C] one-hot masks, both einsums, experts "
                         f"over all E*C capacity slots), 1 x (576+128) tokens = BASELINE config 1 (S={S}, C=528), "
                         f"{len(times)} timed run(s) after 1 warm-up, median {t * 1e3:.1f} ms",
               # the dense form is quadratic in S (3*S^2*M MACs in the two einsums): extrapolated, NOT measured
               "extrapolated_config2_tokens_per_s": None}
    # dense form at the headline's size: the two einsums are 2*S*E*C*M MACs with C ~ S (quadratic), the experts run over
    # E*C rows (linear); shares of the 704-token time taken by FLOP count
    S2 = CONFIGS[2]["B"] * (CONFIGS[2]["n_img"] + CONFIGS[2]["n_txt"])
    f_ein = (2.0 * S * E * 528 * M) / (2.0 * S * E * 528 * M + E * 528 * 3.0 * M * I)
    t2 = t * (f_ein * (S2 / S) ** 2 + (1.0 - f_ein) * (S2 / S))
    literal["extrapolated_config2_tokens_per_s"] = S2 / t2
    literal["extrapolation"] = (f"EXTRAPOLATED, not measured (the masks alone would take 6 GB): time at S = {S} split by FLOP "
                                f"count into einsums ({f_ein:.3f}, scaled by (S2/S1)^2) and experts (scaled by S2/S1), "
                                f"S2 = {S2}: {t2:.1f} s per pass")
    head = index_cfg2 if index_cfg2 is not None else literal
    return {"value": head["tokens_per_s"], "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model, "parity": parity,
            "sample": (f"index-form oracle (restated deepspeed MoE without the dense masks: fp32 gate, top-2 routing with "
                       f"capacity drop, per-expert bf16 MLP over the kept rows, weighted combine; torch CPU) on the headline's "
                       f"own workload, BASELINE config 2: {head['S']} tokens, C = {head.get('capacity')}, "
                       f"{head['runs']} timed pass(es), median {head['seconds']:.2f} s"
                       if index_cfg2 is not None else literal["sample"]),
            "seconds": head["seconds"], "literal_config1": literal}


def self_launch(n: int) -> int:
    """Run this script as n ranks on this node (one per GPU) under torch.distributed.run; returns the exit code.
    Called before anything initialises the GPU in this process, which stays a pure parent (no exec of a GPU process)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 8) // n))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config (per-GPU shape)")
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (default: the config's)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="fp16 = the reference's inference loader dtype")
    ap.add_argument("--decode", type=int, default=0, metavar="S",
                    help="time decode steps of S tokens instead (StableLM block, eval capacity, 12 layers' weights cycled; "
                         "the other_configs decode_S<S> entry on its own, for profiling) and print its summary")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the informational forward+backward timing")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs object")
    args = ap.parse_args()

    if args.decode:
        device = torch.device("cuda", 0)
        torch.cuda.set_device(device)
        print(json.dumps(run_decode(device, torch.cuda.synchronize, (args.decode,), steps=max(args.steps, 120))), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process has not touched the GPU (no HIP call, no library load) and never
        # will -- it starts the N ranks as children of torch.distributed.run, waits, and exits with their code
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    import torch.distributed as dist
    # BENCH_REHEARSAL=1: several ranks share cuda:0 over a gloo group (the all-to-all payload is staged through the
    # host) -- only to rehearse the N > 1 code path on a one-GPU box; the numbers it prints are meaningless.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    default_run = args.config == 2 and not args.batch and args.dtype == "bf16"
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    M, I, E = cfg["M"], cfg["I"], cfg["E"]
    ep = min(world, E)
    while E % ep or world % ep:
        ep -= 1
    ep_rank = rank % ep
    layers = build_layers(cfg, device, ep, ep_rank, dtype=dtype)
    if ep > 1:
        for m in layers:
            m.set_deepspeed_parallelism()
    moe = layers[0]
    x = make_inputs(cfg, cfg["B"], device, seed=1234 + rank, dtype=dtype)
    S = x.shape[0] * x.shape[1]

    from moe_llava_amd import ops

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the headline: W warm-up + K timed steps of the module's default path ----
    elapsed = time_steps(layers, x, args.steps, args.warmup, fence)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    with torch.no_grad():
        out, l_aux, exp_counts = moe(x)
    l_aux, exp_counts = float(l_aux), exp_counts.tolist()

    # ---- roofline of the dominant kernel (fc1 grouped GEMM): a second pass over the same steps with HIP events between
    #      the launches (the timer selects the four-call path: same kernels, same arguments) ----
    timer = ops.PhaseTimer()
    for m in layers:
        m.deepspeed_moe.timer = timer
    time_steps(layers, x, args.steps, 2, fence)
    for m in layers:
        m.deepspeed_moe.timer = None
    fc1_ms = timer.durations("fc1_begin", "fc1_end")[2:]
    fc2_ms = timer.durations("fc1_end", "fc2_end")[2:]
    gate_ms = timer.durations("gate_begin", "gate_end")[2:]
    r = moe.deepspeed_moe.last_routing
    n_kept = float(r.expert_offsets[-1]) / S                 # kept (token, expert) pairs per token on rank 0
    if ep == 1:
        rows_local = int(r.expert_offsets[-1])               # kept (token, expert) pairs = GEMM rows this launch
    else:
        # rows received by this rank's experts per grouped-FFN launch (the exchange is pipelined in chunks)
        rows_local = moe.deepspeed_moe.last_ffn_rows // max(1, moe.deepspeed_moe.last_ffn_launches)
    # what every rank saw: the communicator's world size and the GEMM rows of its experts' launches (N > 1: shows in the
    # record that N ranks met and how the rows were spread)
    ranks_seen, rows_per_gpu = 1, [int(moe.deepspeed_moe.last_ffn_rows) if ep > 1 else int(r.expert_offsets[-1])]
    if world > 1:
        ranks_seen = dist.get_world_size()
        gathered = [None] * world
        dist.all_gather_object(gathered, rows_per_gpu[0])
        rows_per_gpu = [int(v) for v in gathered]
    fc1_avg = statistics.mean(fc1_ms) if fc1_ms else None
    n1 = (2 * I) if cfg["kind"] != "phi" else I
    roofline = None
    if fc1_avg and rows_local is not None:
        flops = 2.0 * rows_local * n1 * M                     # algorithmic: fc1 over kept rows only (rank 0)
        ach = flops / (fc1_avg * 1e-3) / 1e12
        epi = "SWIGLU> (fc1 + SiLU*mul epilogue)" if cfg["kind"] != "phi" else "GELU_NEW> (fc1 + bias + gelu_new epilogue)"
        roofline = {"bound": "mfma", "kernel": "grouped_gemm_kernel<" + epi,
                    "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                    "traffic": None, "avg_launch_ms": fc1_avg, "rows": rows_local,
                    "timing": f"HIP events around the launch on its stream, {len(fc1_ms)} launches, in a pass of their own "
                              f"right after the headline loop",
                    "fc2_avg_launch_ms": statistics.mean(fc2_ms) if fc2_ms else None,
                    "fc2_tflops": (2.0 * rows_local * M * I / (statistics.mean(fc2_ms) * 1e-3) / 1e12) if fc2_ms else None,
                    "gate_routing_avg_ms": statistics.mean(gate_ms) if gate_ms else None}

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the bench, so they come from the
    # committed summary of separate rocprofv3 --pmc passes over this same command (tools/profile_bench.sh ->
    # tools/pmc_summary.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null if absent or if
    # the workload differs from the profiled one.
    if roofline is not None and world == 1 and default_run:
        import glob
        # newest committed summary of THIS config (file name "..._cfg<N>.json") that holds the fc1 kernel
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_summary_*cfg{args.config}.json")),
                       key=lambda f: (os.path.basename(os.path.dirname(f)), os.path.basename(f)), reverse=True)

        def is_fc1(name):                                    # grouped_gemm_kernel<DT, EPI, BM, BNW>: bf16 SwiGLU 256 x 256
            if "grouped_gemm_kernel<" not in name:
                return False
            targs = [t.strip() for t in name.split("<", 1)[1].split(">", 1)[0].split(",")]
            return targs[-3:] == ["1", "256", "256"] and (len(targs) == 3 or targs[0] == "0")
        from moe_llava_amd.build import kernel_source_sha
        sha_now = kernel_source_sha()
        for pmc in cands:
            summary = json.load(open(pmc))
            k = next((v for name, v in summary.items() if is_fc1(name) and "hbm_bytes_per_launch" in v), None)
            if k is None:
                continue
            if summary.get("_meta", {}).get("kernel_source_sha256") != sha_now:
                # the newest summary was taken on other kernel sources: no number rather than a stale one
                roofline["traffic_unit"] = ("null: " + os.path.relpath(pmc, ROOT) + " was profiled on other kernel sources "
                                            "(re-run tools/profile_bench.sh)")
                break
            roofline["traffic"] = k["hbm_bytes_per_launch"]
            roofline["traffic_unit"] = ("bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc, "
                                        + os.path.relpath(pmc, ROOT) + ")")
            roofline["algorithmic_bytes"] = 2.0 * (2 * I * M * E) + 2.0 * S * M + 2.0 * rows_local * I
            break
    # ---- informational: one training step (forward + backward through the autograd node) on the same batch; not part
    #      of `value` (BASELINE's metric is the forward block) ----
    training = None
    if world == 1 and not args.no_train and args.dtype == "bf16" and args.config in (1, 2, 4):
        moe.train()
        R = torch.randn_like(x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        fw, bw = [], []
        for i in range(2 + 5):
            moe.zero_grad(set_to_none=True)
            xg = x.detach().requires_grad_(True)
            ev[0].record()
            o, la, _ = moe(xg)
            ev[1].record()
            torch.autograd.backward([o, la], [R, torch.ones_like(la) * 0.01])
            ev[2].record()
            torch.cuda.synchronize()
            if i >= 2:
                fw.append(ev[0].elapsed_time(ev[1]))
                bw.append(ev[1].elapsed_time(ev[2]))
        moe.zero_grad(set_to_none=True)
        f, b = statistics.median(fw), statistics.median(bw)
        training = {"fwd_ms": f, "bwd_ms": b, "tokens_per_s": S / (f + b) * 1e3,
                    "note": "forward (keeping pre-activations) + backward of the block, 5 steps, HIP events"}
        del R, xg, o, la
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline and default_run:
            state = {k: v.detach().cpu() for k, v in moe.state_dict().items()}
            cpu = cpu_baseline(state, gpu_moe=moe.train(), device=device)
        ms = elapsed / args.steps * 1e3
        others = None
        if world == 1 and default_run and not args.no_other_configs:
            del layers, moe, x, out, r
            torch.cuda.empty_cache()
            others = run_other_configs(device, fence)
        line = {
            "metric": f"MoE-layer tokens/sec (top-2, {E} experts, {cfg['kind']} block)",
            "value": world * S * args.steps / elapsed,
            "unit": "tokens/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "rows_per_gpu": rows_per_gpu,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic" + (" (REHEARSAL: ranks share one GPU over gloo, timing meaningless)" if rehearsal else ""),
            "config": {"workload": f"{cfg['tag']}: d={M}, expert width {I} ({cfg['kind']}), {E} experts top-{TOPK}, capacity "
                                   f"factor {cfg['cf']} (training capacity), 2nd-expert Gumbel sampling on (upstream "
                                   f"default), {cfg['B']} x ({cfg['n_img']} img + {cfg['n_txt']} text) tokens per GPU, "
                                   f"{cfg['layers']} layer(s) of weights cycled",
                       "tokens_per_gpu": S, "global_tokens": world * S,
                       "parallelism": "single GPU" if world == 1 else f"ep{ep} x dp{world // ep} (RCCL all-to-all)",
                       "kept_pairs_per_token": n_kept,
                       "path": "default module path: one moe_block_top2_fwd call per step" if world == 1 else
                               "expert-parallel exchange (variable-split all_to_all_single of kept rows)"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "training": training,
            "layer_tflops": (n_kept * flops_per_row(cfg) * S / (ms * 1e-3) / 1e12) if n_kept else None,
            "l_aux": l_aux, "exp_counts": exp_counts,
            "other_configs": others,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
