#!/usr/bin/env python3
"""bench.py — headline benchmark of the VITS hot path on MI355X (contract: see the task statement / DESIGN.md §6).

Metric (BASELINE.json): audio samples/s end-to-end (+ RTF) on "vits-english, batch=64 fixed-length 128-phoneme
utterances": one STEP = one pass of the whole path (text encoder -> stochastic duration predictor -> alignment ->
coupling flow -> HiFiGAN) over one batch of synthetic utterances, ids in (host, 32 KB) -> fp32 PCM resident in HBM.
Weights are the deterministic synthetic MMS-TTS-architecture model (no checkpoint is available offline), ids and noise are
the counter-based synthetic streams of include/vits_synth_noise.h. Nothing is skipped in the timed region; durations are
the model's own predictions (data-dependent shapes, one host read of B frame counts per step like the reference's
vits.cpp:1133), unless --pinned is given.

N GPUs: one process per GPU (torch.distributed, backend nccl == RCCL), rank r synthesises its own 64 utterances
(weak scaling, no data-path collective) and the PCM of all ranks is all-gathered over xGMI at the end of every step.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA / vector peak
PEAK_HBM_GBS = 8000.0


def load_package():
    name = "vits_cpp_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "vits.cpp_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def algorithmic_flops(T, frames):
    """SURVEY.md §8(d) / BASELINE.md §4 FLOP model for one utterance of T ids and L frames (HF-crop sizes)."""
    L = frames
    enc = 2 * (6 * (1035648 * T + 384 * T * T) + 73728 * T)
    dur = 1.08e6 * T
    return enc + dur + 14155776 * L + 614907904 * L


def cpu_baseline(model_bytes, ids, seed, budget_s=15.0):
    """Times the CPU oracle (restatement of the reference ggml graph; the ggml fork itself is not vendored) on this host's
    cores with the reference's method (sequential batch-1 calls, wall clock; test/bench_e2e.cpp:79-89) on a bounded sample
    of the same workload. Thread count: the reference would use max(hardware_concurrency, 6) (src/include/common.h:19-21),
    which on a 256-thread host is far past the oracle's scaling knee, so a short sweep picks the FASTEST thread count and that
    one is reported (`cores`); the figure at the reference's rule is kept alongside for transparency."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    m = O.Model(model_bytes)
    hw = max(os.cpu_count() or 1, 6)

    def run(u, threads):
        t = time.perf_counter()
        r = m.process_ids(ids[u], mode=O.MODE_REFERENCE, noise_kind=O.NOISE_COUNTER, noise_seed=seed + u, threads=threads, taps=["waveform"])
        return r["waveform"].size, time.perf_counter() - t

    sweep = {}
    for th in sorted({min(hw, c) for c in (8, 16, 32, 64)} | {hw}):
        n, dt = run(0, th)
        sweep[th] = n / dt
    best = max(sweep, key=sweep.get)
    t0 = time.perf_counter()
    samples, n = 0, 0
    for u in range(ids.shape[0]):
        sz, _ = run(u, best)
        samples += sz
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": samples / dt, "unit": "samples/s", "cores": best, "kind": "port",
            "sample": f"{n} utterance(s) of the same workload ({ids.shape[1]} ids each, reference mode), {dt:.1f} s wall, "
                      f"CPU restatement of the reference ggml graph (ggml fork not vendored); thread count = fastest of a sweep",
            "rtf_16k": dt / (samples / 16000.0), "host_threads": hw, "thread_sweep_samples_per_s": {str(k): v for k, v in sweep.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--ids-per-utt", type=int, default=128, help="encoder input ids per utterance (after blank interspersing)")
    ap.add_argument("--pinned", type=int, default=0, help=">0: pin every id to this many frames (SURVEY §8d run ii)")
    ap.add_argument("--mode", choices=["reference", "hf"], default="reference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--pcm16", action="store_true", help="multi-GPU: convert to int16 on the device and gather that (half the bytes)")
    ap.add_argument("--chunk-frames", type=int, default=0,
                    help="run the vocoder in windows of this many frames (vits_process_opts.vocoder_chunk_frames); 0 = whole utterance")
    ap.add_argument("--single-pass", action="store_true",
                    help="instrumented (serialised) warmup + timed region only, no second pass: every launch of the run is then one "
                         "the HIP events timed, which is what a rocprofv3 --kernel-trace of this command is compared against")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    # VITS_BENCH_FORCE_DIST=1: run the RCCL exchange even with one rank (exercises the N > 1 code path on a 1-GPU box)
    dist_on = world > 1 or os.environ.get("VITS_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    pkg = load_package()
    pkg.set_device(local_rank)
    spec = importlib.util.spec_from_file_location("vits_multi_gpu", os.path.join(ROOT, "vits.cpp_amd", "multi_gpu.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)

    B, T = args.batch, args.ids_per_utt
    mode = pkg.MODE_REFERENCE if args.mode == "reference" else pkg.MODE_HF
    model_bytes = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL)
    model = pkg.Model(model_bytes)
    model.set_mode(mode)
    # rank r owns utterances [r*B, (r+1)*B): ids seed 1234+utt, noise seed 4321+utt
    ids = pkg.synth_ids(B * world, T)[rank * B:(rank + 1) * B]
    noise_seed = 4321 + rank * B
    cap = 256 * 8 * T + 294  # PCM row capacity: up to 8 frames per id
    out = torch.empty((B, cap), dtype=torch.float32, device="cuda")

    def step(profile=False):
        _, lengths, frames = model.process_batch(ids, mode=mode, noise_kind=pkg.NOISE_COUNTER, noise_seed=noise_seed, fixed_duration=args.pinned,
                                                 out_device=out.data_ptr(), out_device_stride=cap, skip_host_copy=True,
                                                 vocoder_chunk_frames=args.chunk_frames)
        if dist_on:
            # the path's only exchange: ragged all-gather of the PCM (lengths first) over RCCL/xGMI
            lens_d = torch.from_numpy(lengths).cuda()
            mg.gather_pcm(mg.to_pcm16(pkg, out, lens_d) if args.pcm16 else out, lens_d)
        return lengths, frames

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if args.single_pass and not args.no_prof:
        model.prof_enable(True)
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_prof:
        model.prof_reset()
        model.prof_enable(True)
    t0 = time.perf_counter()
    total_samples = 0
    for _ in range(args.steps):
        lengths, frames = step()
        total_samples += int(lengths.sum())
    fence()
    elapsed = time.perf_counter() - t0
    model.prof_enable(False)
    # second pass of the same K steps in the library's default configuration: no per-kernel events, and therefore the three
    # resblocks of every vocoder stage on concurrent streams (the profiler serialises them so that kernel durations are
    # meaningful). Reported beside the headline, which stays the instrumented region the roofline is measured in.
    elapsed_plain = None
    if not args.no_prof and not args.single_pass:
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed_plain = time.perf_counter() - t1
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ts = torch.tensor([total_samples], dtype=torch.int64, device="cuda")
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_samples = int(ts.item())

    if rank == 0:
        sr = model.sampling_rate
        value = total_samples / elapsed
        flops_step = sum(algorithmic_flops(T, int(f)) for f in frames) * world
        res = {
            "metric": "audio samples/sec end-to-end (ids -> fp32 PCM in HBM), vits-english architecture, batch=64 x 128 ids",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"vits-english (MMS-TTS architecture, synthetic weights), batch={B} per GPU, {T} ids per utterance, "
                                   f"{'predicted' if not args.pinned else 'pinned %d frames/id' % args.pinned} durations, mode={args.mode}",
                       "batch_per_gpu": B, "ids_per_utterance": T, "frames_per_utterance_mean": float(np.mean(frames)),
                       "samples_per_step": total_samples // args.steps, "sampling_rate": sr, "parallelism": f"utterance-sharded x{world}",
                       "pcm_destination": "device (HBM)", "vocoder_chunk_frames": args.chunk_frames},
            "rtf": elapsed / (total_samples / float(sr)), "rtf_22050": elapsed / (total_samples / 22050.0),
            "algorithmic_tflops": flops_step * args.steps / elapsed / 1e12,
            "frac_fp32_peak_whole_path": flops_step * args.steps / elapsed / 1e12 / (PEAK_F32_TFLOPS * world),
        }
        if elapsed_plain is not None and world == 1:
            res["value_without_kernel_events"] = total_samples / elapsed_plain
            res["ms_per_step_without_kernel_events"] = 1000.0 * elapsed_plain / args.steps
            res["without_kernel_events_note"] = "library default: no per-kernel HIP events, resblocks of a stage on 3 concurrent streams" 
        if not args.no_prof:
            rep = model.prof_report()["kernels"]
            # group by kernel instantiation (taps, tile, epilogue) == one rocprofv3 kernel name
            groups = {}
            for k in rep:
                parts = k["name"].split("|")
                key = "|".join(parts[1:5]) if len(parts) >= 5 else k["name"]
                g = groups.setdefault(key, {"calls": 0, "ms": 0.0, "flop": 0.0, "bytes": 0.0, "labels": set()})
                g["calls"] += k["calls"]
                g["ms"] += k["ms"]
                g["flop"] += k["flop"]
                g["bytes"] += k["bytes"]
                g["labels"].add(parts[0])
            dom_key = max(groups, key=lambda kk: groups[kk]["ms"])
            dom = groups[dom_key]
            avg_ms = dom["ms"] / dom["calls"]
            achieved = dom["flop"] / dom["calls"] / (avg_ms * 1e-3) / 1e12
            all_ms = sum(g["ms"] for g in groups.values())
            res["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_TFLOPS,
                               "traffic": None, "kernel": "conv_mfma_kernel<" + dom_key + ">", "avg_launch_ms": avg_ms, "launches": dom["calls"],
                               "share_of_gpu_time": dom["ms"] / all_ms,
                               "algorithmic_gbytes_per_launch": dom["bytes"] / dom["calls"] / 1e9,
                               "hbm_frac_if_algorithmic": dom["bytes"] / dom["calls"] / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
            # HBM bytes per launch of that kernel from the PMC passes (collected separately with rocprofv3 --pmc on this
            # same command; PMC cannot be sampled from inside the run): profiles/round1_v12_pmc_traffic.json
            try:
                tiles = {"t0": "2, 2, 2, 2", "t1": "1, 4, 2, 2", "t2": "1, 4, 1, 2", "t3": "1, 4, 2, 1", "t4": "1, 4, 1, 1"}
                kk, dd, tt, ee = dom_key.split("|")
                with open(os.path.join(ROOT, "profiles", "round1_v12_pmc_traffic.json")) as fh:
                    pmc = json.load(fh)["kernels"]
                cands = [v for n, v in pmc.items() if n.startswith(f"void vits::conv_mfma_kernel<{kk[1:]}, {dd[1:]}, ")
                         and n.endswith(f"{tiles[tt]}, {ee[1:]}>(vits::ConvParams)")]
                if cands:
                    best = max(cands, key=lambda v: v["launches_sampled"])
                    res["roofline"]["traffic"] = best["hbm_bytes_per_launch"]
                    res["roofline"]["traffic_unit"] = "bytes per launch (PMC FETCH_SIZE calibrated x1.143 + WRITE_SIZE, profiles/round1_v12_pmc_traffic.json)"
                    res["roofline"]["traffic_over_algorithmic"] = best["hbm_bytes_per_launch"] / (dom["bytes"] / dom["calls"])
            except Exception:
                pass
            conv_ms = sum(g["ms"] for kk, g in groups.items() if kk.startswith("k"))
            conv_flop = sum(g["flop"] for kk, g in groups.items() if kk.startswith("k"))
            res["kernel_time_ms_per_step"] = all_ms / args.steps
            res["all_conv_kernels"] = {"tflops": conv_flop / (conv_ms * 1e-3) / 1e12, "frac_of_peak": conv_flop / (conv_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                                       "share_of_gpu_time": conv_ms / all_ms}
            top = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:8]
            res["top_kernels"] = [{"kernel": kk, "ms_per_step": g["ms"] / args.steps, "calls_per_step": g["calls"] / args.steps,
                                   "tflops": (g["flop"] / (g["ms"] * 1e-3) / 1e12) if g["flop"] else None} for kk, g in top]
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(model_bytes, ids, noise_seed)
            res["cpu_baseline"] = cb
            res["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(res))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    model.close()


if __name__ == "__main__":
    main()
