#!/usr/bin/env python3
"""MoE-layer throughput bench (BASELINE.json metric: MoE-layer tokens/s on MI355X, top-2, 4 experts).

A "step" is one forward of the sparse-MoE FFN block (gate -> top-2 routing with capacity drop -> grouped expert
FFN -> weighted combine) over one synthetic batch that is already resident in HBM.

  N = 1 : BASELINE config 2 -- MoE-LLaVA-StableLM-1.6B-4e block (d=2048, expert width 5632 SwiGLU, 4 experts,
          top-2, capacity factor 1.5, training-mode capacity), batch 32 x (576 image + 128 text) tokens, bf16.
  N > 1 : the same block and per-GPU batch on every rank (weak scaling); the 4 experts shard expert-parallel
          over min(N, 4) consecutive ranks (x N/4 replicas), token dispatch / combine by RCCL all-to-all.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel (the
fc1 grouped GEMM, timed with HIP events inside the timed region) and `cpu_baseline` (the literal dense-mask
oracle of the reference layer on the host cores, on one 704-token sample = BASELINE config 1).
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

M, I, E, TOPK, CF = 2048, 5632, 4, 2, 1.5
N_IMG, N_TXT, BATCH = 576, 128, 32
PEAK_BF16_TFLOPS = 2500.0     # dense MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def make_inputs(batch: int, device, seed: int):
    """text rows ~ N(0,1); image rows N(0,1) + 0.5*u_img (one fixed unit direction, gives the
    modality-skewed routing); bf16."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch, N_IMG + N_TXT, M, generator=g)
    u = torch.randn(M, generator=torch.Generator().manual_seed(4321))
    u = u / u.norm() * (M ** 0.5)
    x[:, :N_IMG] += 0.5 * u
    return x.to(torch.bfloat16).to(device)


def build_block(device, ep_size: int, seed: int = 1234):
    import moe_llava_amd
    from moe_llava_amd.expert_mlps import StableLMMLP
    torch.manual_seed(seed)
    proto = StableLMMLP(M, I).to(torch.bfloat16)
    # the nine kwargs of the reference's constructor call (llava_stablelm_moe.py:550-560, finetune_moe.sh:3-13)
    moe = moe_llava_amd.MoE(M, expert=proto, num_experts=E, ep_size=ep_size, k=TOPK, capacity_factor=CF,
                            eval_capacity_factor=2.0, min_capacity=0, use_residual=False)
    g = torch.Generator().manual_seed(seed)
    state = {"wg": torch.randn(E, M, generator=g) * 0.02}
    experts = []
    for e in range(E):
        experts.append({n: (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
                        for n, shape in (("gate_proj", (I, M)), ("up_proj", (I, M)), ("down_proj", (M, I)))})
    return moe, state, experts


def load_weights(moe, state, experts, device, ep_rank: int):
    el = moe.num_local_experts
    moe.deepspeed_moe.gate.wg.weight.data = state["wg"].clone()
    for le, mod in enumerate(moe.deepspeed_moe.experts.deepspeed_experts):
        w = experts[ep_rank * el + le]
        mod.gate_proj.weight.data = w["gate_proj"].clone()
        mod.up_proj.weight.data = w["up_proj"].clone()
        mod.down_proj.weight.data = w["down_proj"].clone()
    return moe.to(device).train()


def cpu_baseline(state, experts, budget_s: float = 12.0, gpu_moe=None, device=None):
    """Literal dense-mask restatement of the reference layer (oracle) on the host cores, one 704-token sample
    (BASELINE config 1: S=704, C=528).  Checker code timed as a reported baseline, never shipped."""
    from oracle import ds_moe_literal as lit
    from oracle.expert_mlps import StableLMMLPRef
    torch.manual_seed(0)
    ref = lit.MoERef(M, StableLMMLPRef(M, I).to(torch.bfloat16), num_experts=E, ep_size=1, k=TOPK,
                     capacity_factor=CF, eval_capacity_factor=2.0, min_capacity=0, use_residual=False)
    ref.deepspeed_moe.gate.wg.weight.data = state["wg"].clone()
    for e, mod in enumerate(ref.deepspeed_moe.experts.deepspeed_experts):
        mod.gate_proj.weight.data = experts[e]["gate_proj"].clone()
        mod.up_proj.weight.data = experts[e]["up_proj"].clone()
        mod.down_proj.weight.data = experts[e]["down_proj"].clone()
    ref.train()
    x = make_inputs(1, "cpu", seed=1234)
    S = x.shape[0] * x.shape[1]
    ref.deepspeed_moe.gate.noise = lit.gumbel_noise((S, E), 99)    # upstream default: 2nd-expert sampling on
    times = []
    t_all = time.perf_counter()
    with torch.no_grad():
        ref(x)                                               # warm-up (thread pool, allocator)
        t_all = time.perf_counter()
        while time.perf_counter() - t_all < budget_s and len(times) < 2000:
            t0 = time.perf_counter()
            ref(x)
            times.append(time.perf_counter() - t0)
    t = statistics.median(times)
    timed = times
    times = [None] + times                                   # (one warm-up run, not timed)
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
            parity = {"sample": f"S={S}, C={r.capacity}, oracle fed the kernel's logits and the same noise",
                      "routing_bit_exact": bool(torch.equal(r.expert_idx.cpu(), want["expert_idx"])
                                                and torch.equal(r.slot.cpu(), want["slot"])
                                                and torch.equal(g_cnt.cpu(), r_cnt)
                                                and torch.equal(r.kept_counts.cpu(), want["kept_counts"])),
                      "l_aux_abs_err": abs(float(g_laux) - float(r_laux)),
                      "out_max_abs_err": float(d.max()), "out_ref_max_abs": float(r_out.float().abs().max()),
                      "out_within_bar": bool((d <= 4e-3 * float(r_out.float().abs().max())
                                              + 1e-2 * r_out.float().abs()).all()),
                      "out_bar": "|d| <= 4e-3 * max|ref| + 1e-2 * |ref| (tests/_parity.py)",
                      "tie_sensitive_tokens": int(((top2[:, 0] - top2[:, 1]) < 1e-5).sum())}
        finally:
            gate.noise_override = None
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": S / t, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model, "parity": parity,
            "sample": f"literal dense-mask oracle (restated deepspeed MoE, torch CPU bf16), 1 x ({N_IMG}+{N_TXT}) tokens "
                      f"= BASELINE config 1 (S={S}, C=528; the dense [S,E,C] masks of the full 22528-token batch would "
                      f"take 6 GB), repeated for {sum(timed):.1f} s of CPU work: {len(timed)} timed run(s) after "
                      f"{len(times) - len(timed)} warm-up, median {t * 1e3:.1f} ms",
            "seconds": t}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="samples per GPU (704 tokens each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the informational forward+backward timing")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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

    ep = min(world, E)
    while E % ep or world % ep:
        ep -= 1
    ep_rank = rank % ep
    moe, state, experts = build_block(device, ep)
    moe = load_weights(moe, state, experts, device, ep_rank)
    if ep > 1:
        moe.set_deepspeed_parallelism()
    x = make_inputs(args.batch, device, seed=1234 + rank)
    S = x.shape[0] * x.shape[1]

    from moe_llava_amd import ops
    timer = ops.PhaseTimer()

    def step():
        with torch.no_grad():
            return moe(x)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    moe.deepspeed_moe.timer = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, l_aux, exp_counts = step()
    fence()
    elapsed = time.perf_counter() - t0
    moe.deepspeed_moe.timer = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)

    # ---- roofline of the dominant kernel (fc1 grouped GEMM), HIP-event timed inside the timed region ----
    fc1_ms = timer.durations("fc1_begin", "fc1_end")
    fc2_ms = timer.durations("fc1_end", "fc2_end")
    gate_ms = timer.durations("gate_begin", "gate_end")
    r = moe.deepspeed_moe.last_routing
    if ep == 1:
        rows_local = int(r.expert_offsets[-1])               # kept (token, expert) pairs = GEMM rows this launch
    else:
        # rows received by this rank's experts per grouped-FFN launch (the exchange is pipelined in chunks)
        rows_local = moe.deepspeed_moe.last_ffn_rows // max(1, moe.deepspeed_moe.last_ffn_launches)
    fc1_avg = statistics.mean(fc1_ms) if fc1_ms else None
    roofline = None
    if fc1_avg and rows_local is not None:
        flops = 2.0 * rows_local * (2 * I) * M                # algorithmic: gate+up projections over kept rows only (rank 0)
        ach = flops / (fc1_avg * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "grouped_gemm_kernel<SWIGLU> (fc1 + SiLU*mul epilogue)",
                    "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                    "traffic": None, "avg_launch_ms": fc1_avg, "rows": rows_local,
                    "fc2_avg_launch_ms": statistics.mean(fc2_ms) if fc2_ms else None,
                    "fc2_tflops": (2.0 * rows_local * M * I / (statistics.mean(fc2_ms) * 1e-3) / 1e12) if fc2_ms else None,
                    "gate_routing_avg_ms": statistics.mean(gate_ms) if gate_ms else None}

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the bench, so they come from the
    # committed summary of separate rocprofv3 --pmc passes over this same command (tools/profile_bench.sh ->
    # tools/pmc_summary.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null if absent or if
    # the workload differs from the profiled one.
    if roofline is not None and world == 1 and args.batch == BATCH:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary_*.json")))
        if cands:
            pmc = cands[-1]                                  # newest committed summary
            summary = json.load(open(pmc))
            k = next((v for name, v in summary.items() if name.startswith("void moe::grouped_gemm_kernel<1")), {})
            if "hbm_bytes_per_launch" in k:
                roofline["traffic"] = k["hbm_bytes_per_launch"]
                roofline["traffic_unit"] = ("bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc, "
                                            + os.path.relpath(pmc, ROOT) + ")")
                roofline["algorithmic_bytes"] = 2.0 * (2 * I * M * E) + 2.0 * S * M + 2.0 * rows_local * I
    # ---- informational: one training step (forward + backward through the autograd node) on the same batch; not part
    #      of `value` (BASELINE's metric is the forward block) ----
    training = None
    if world == 1 and not args.no_train:
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
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(state, experts, gpu_moe=moe.train(), device=device)
        ms = elapsed / args.steps * 1e3
        n_kept = float(r.expert_offsets[-1]) / S             # kept (token, expert) pairs per token on rank 0
        line = {
            "metric": "MoE-layer tokens/sec (top-2, 4 experts, StableLM-1.6B-4e block)",
            "value": world * S * args.steps / elapsed,
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU over gloo, timing meaningless)" if rehearsal else ""),
            "config": {"workload": f"BASELINE config 2: StableLM-1.6B-4e MoE block forward, d={M}, expert width {I} SwiGLU, "
                                   f"{E} experts top-{TOPK}, capacity factor {CF} (training capacity), 2nd-expert Gumbel "
                                   f"sampling on (upstream default), {args.batch} x ({N_IMG} img + {N_TXT} text) tokens per GPU",
                       "tokens_per_gpu": S, "global_tokens": world * S,
                       "parallelism": "single GPU" if world == 1 else f"ep{ep} x dp{world // ep} (RCCL all-to-all)",
                       "kept_pairs_per_token": n_kept},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "training": training,
            "layer_tflops": (n_kept * 6.0 * M * I * S / (ms * 1e-3) / 1e12) if n_kept else None,
            "l_aux": float(l_aux), "exp_counts": exp_counts.tolist(),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
