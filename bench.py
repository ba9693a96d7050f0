#!/usr/bin/env python3
"""bench.py — headline benchmark of the VITS hot path on MI355X (contract: see the task statement / DESIGN.md §6).

Metric (BASELINE.json): audio samples/s end-to-end (+ RTF) on "vits-english, batch=64 fixed-length 128-phoneme
utterances": one STEP = one pass of the whole path (text encoder -> stochastic duration predictor -> alignment ->
coupling flow -> HiFiGAN) over one batch of synthetic utterances, ids in (host, 32 KB) -> fp32 PCM resident in HBM.
Weights are the deterministic synthetic MMS-TTS-architecture model (no checkpoint is available offline), ids and noise are
the counter-based synthetic streams of include/vits_synth_noise.h. Nothing is skipped in the timed region; durations are
the model's own predictions (data-dependent shapes, one host read of B frame counts per step like the reference's
vits.cpp:1133), unless --pinned is given.

N GPUs (`--gpus N`): one process per GPU (torch.distributed, backend nccl == RCCL), rank r synthesises its own 64
utterances (weak scaling, no data-path collective) and the PCM of all ranks is all-gathered over xGMI at the end of every
step. Started by torchrun (RANK/LOCAL_RANK/WORLD_SIZE in the environment) this process IS one rank; started plainly with
--gpus N > 1 it is a LAUNCHER: it starts the N rank processes itself before touching the GPU, relays rank 0's JSON
line, and fails if any rank fails or fewer than N devices are visible — it never falls back to fewer ranks.

Prints ONE JSON line (< 4 KB: compact_line) on rank 0, LAST on stdout; the full record goes to bench_detail.json.
"""
import argparse
import gc
import glob
import importlib.util
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA / vector peak
PEAK_HBM_GBS = 8000.0
PEAK_MFMA16_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md; AMD's headline figure has 2:1 sparsity)
ALG_BYTES_PER_SAMPLE_F32 = 19.6e3 + 0.1e3  # layer-granular fp32 activation model (HiFiGAN + flow), per output sample
# the same layer-granular model with 16-bit conv inputs: a resblock conv pair moves 16 B per element (fp32: 24 B),
# the transposed convs and conv_pre / conv_post read 16-bit inputs -> 0.66 of the fp32 figure
ALG_BYTES_PER_SAMPLE_16 = ALG_BYTES_PER_SAMPLE_F32 * 16.0 / 24.0


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


def cpu_baseline(jobs, mode_name, with_one_thread=True, budget_s=9.0):
    """Times the CPU oracle (restatement of the reference ggml graph; the ggml fork itself is not vendored) on this host's
    cores with the reference's method (sequential batch-1 calls, wall clock; test/bench_e2e.cpp:79-89) on a bounded sample
    of the same workload. Three SUSTAINED figures (each after one untimed warm-up call: the first call pages the model in):
      * at the reference's thread rule max(hardware_concurrency, 6) (src/include/common.h:19-21),
      * at the fastest thread count of a short sweep (on a 256-thread host the rule is far past the oracle's scaling knee) —
        this one is `value` / `cores`,
      * at 1 thread.
    jobs: [(model_bytes, ids [n, T], noise_seed)]; utterances are taken round-robin from the jobs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    models = [O.Model(mb) for mb, _, _ in jobs]
    omode = O.MODE_REFERENCE if mode_name == "reference" else O.MODE_HF
    hw = max(os.cpu_count() or 1, 6)
    order = [(j, u) for u in range(max(len(i) for _, i, _ in jobs)) for j in range(len(jobs)) if u < len(jobs[j][1])]

    def run(k, threads):
        j, u = order[k % len(order)]
        t = time.perf_counter()
        r = models[j].process_ids(jobs[j][1][u], mode=omode, noise_kind=O.NOISE_COUNTER, noise_seed=jobs[j][2] + u, threads=threads, taps=["waveform"])
        return r["waveform"].size, time.perf_counter() - t

    def sustained(threads, budget, max_n):
        t0 = time.perf_counter()
        samples = n = 0
        while n < max_n:
            sz, _ = run(n, threads)
            samples += sz
            n += 1
            if time.perf_counter() - t0 > budget:
                break
        dt = time.perf_counter() - t0
        return {"value": samples / dt, "threads": threads, "utterances": n, "wall_s": dt, "rtf_16k": dt / (samples / 16000.0)}

    run(0, min(hw, 32))  # warm-up (untimed)
    sweep = {}
    for th in sorted({min(hw, c) for c in (8, 16, 32, 64)}):
        n, dt = run(0, th)
        sweep[th] = n / dt
    best = max(sweep, key=sweep.get)
    fastest = sustained(best, budget_s, len(order))
    at_rule = fastest if hw == best else sustained(hw, 4.0, 4)
    one = sustained(1, 1e9, 3) if with_one_thread else None  # three utterances (several seconds each)
    T = jobs[0][1].shape[1]
    cpu = host_cpu_info()
    # `cores` = the PHYSICAL cores of the host the baseline ran on (what the north star asks to be stated); `threads` = the oracle threads of the
    # fastest configuration of the sweep, which is what `value` was measured with (round 5 printed that thread count as "cores")
    res = {"value": fastest["value"], "unit": "samples/s", "cores": cpu["physical_cores"], "threads": best, "cpu_model": cpu["cpu_model"], "kind": "port",
           "sample": f"{fastest['utterances']} utterance(s) of the same workload ({T} ids each, {mode_name} mode), {fastest['wall_s']:.1f} s wall, sequential "
                     f"batch-1 calls (method of test/bench_e2e.cpp:79-89), CPU restatement of the reference ggml graph (ggml fork not vendored); "
                     f"thread count = fastest of the sweep {sorted(sweep)}",
           "sample_short": f"{fastest['utterances']} utterances x {T} ids of the same workload, {fastest['wall_s']:.1f} s, sequential batch-1 calls, C++ oracle (port of the ggml graph)",
           "rtf_16k": fastest["rtf_16k"], "host_threads": os.cpu_count(),
           "at_reference_thread_rule": at_rule, "thread_sweep_samples_per_s": {str(k): v for k, v in sweep.items()}}
    if one:
        res["one_thread"] = one
    return res


def host_cpu_info():
    """CPU model, physical core count and hardware threads of this host from /proc/cpuinfo ("print CPU model and core count"; the north
    star: "core count stated"). Physical cores = distinct (physical id, core id) pairs; falls back to the thread count where the file lacks them."""
    model, pairs, phys, core, threads = None, set(), None, None, 0
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                k, _, v = ln.partition(":")
                k, v = k.strip(), v.strip()
                if k == "processor":
                    threads += 1
                    phys = core = None
                elif k == "model name" and model is None:
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    threads = threads or (os.cpu_count() or 1)
    return {"cpu_model": model or "unknown", "physical_cores": len(pairs) or threads, "hardware_threads": threads}


def duration_boundary_margin(pkg, model, model_bytes, ids, noise_base, mode, mode_name):
    """Error bars on "durations bit-exact vs the real ggml path". A duration is ceil(exp(logw) * length_scale) (vits.cpp:996-1001).
    Over every id of the benchmark batch, from the ORACLE's stage one (vo_log_durations), checked against the GPU's durations:
      * how many w = exp(logw) lie within 1e-4 / 1e-3 / 1e-2 (relative) of a ceil() boundary (what an implementation whose log-durations
        differ by eps could move at most);
      * Q6: how many duration-predictor latents lie OUTSIDE the spline interval [-5, 5] — where the reference's masked get / set pair
        misaligns (vits.cpp:832-849; reproduced literally in reference mode since round 4). 0 means Q6 plays no part in this batch;
      * Q8, MEASURED instead of bounded: the same stage one with EMULATED ggml lookup tables (tanh-GELU and the soft-max exponential
        through fp16 tables, double sum — inferred from upstream ggml, the reference's fork is absent): how many of the durations change
        against the erf-GELU / fp32-soft-max reading. In that mode both sides compute stage one in ONE shared order of operations
        (include/vits_exact_math.h; vits_model_set_ggml_tables(model, 1) / vo_opts.ggml_tables = 1), so GPU and oracle must agree on EVERY
        duration and every log-duration bit (`gpu_vs_oracle_durations_differ` = 0, `max_abs_log_duration_gpu_minus_oracle` = 0; round 4, with
        the tables inside the throughput kernels: 5 of 8,192)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    om = O.Model(model_bytes)
    omode = O.MODE_REFERENCE if mode_name == "reference" else O.MODE_HF
    t0 = time.perf_counter()
    th = min(os.cpu_count() or 1, 16)
    model.process_batch(ids, mode=mode, noise_kind=pkg.NOISE_COUNTER, noise_seed=noise_base, collect_taps=True, frames_only=True, keep_pcm=False)
    g_dur = [model.tap("durations", u).copy() for u in range(len(ids))]
    g_logw = [model.tap("log_duration", u).copy() for u in range(len(ids))]
    model.set_ggml_tables(True)
    try:
        model.process_batch(ids, mode=mode, noise_kind=pkg.NOISE_COUNTER, noise_seed=noise_base, collect_taps=True, frames_only=True, keep_pcm=False)
        gt_dur = [model.tap("durations", u).copy() for u in range(len(ids))]
        gt_logw = [model.tap("log_duration", u).copy() for u in range(len(ids))]
    finally:
        model.set_ggml_tables(False)
    rel, equal, total, maxdev, outside = [], 0, 0, 0.0, 0
    tab = {"oracle_durations_changed": 0, "gpu_durations_changed": 0, "gpu_vs_oracle_durations_differ": 0, "max_abs_log_duration_change": 0.0,
           "max_abs_log_duration_gpu_minus_oracle": 0.0}
    for u in range(len(ids)):
        logw, dur = om.log_durations(ids[u], mode=omode, noise_kind=O.NOISE_COUNTER, noise_seed=noise_base + u, threads=th)
        outside += O.outside_latents()
        w = np.exp(logw.astype(np.float64))
        rel.append(np.abs(w - np.round(w)) / np.maximum(w, 1e-30))
        equal += int((g_dur[u] == dur).sum())
        total += dur.size
        maxdev = max(maxdev, float(np.abs(g_logw[u] - logw).max()))
        logw_t, dur_t = om.log_durations(ids[u], mode=omode, noise_kind=O.NOISE_COUNTER, noise_seed=noise_base + u, threads=th, ggml_tables=True)
        tab["oracle_durations_changed"] += int((dur_t != dur).sum())
        tab["gpu_durations_changed"] += int((gt_dur[u] != g_dur[u]).sum())
        tab["gpu_vs_oracle_durations_differ"] += int((gt_dur[u] != dur_t).sum())
        tab["max_abs_log_duration_change"] = max(tab["max_abs_log_duration_change"], float(np.abs(logw_t - logw).max()))
        tab["max_abs_log_duration_gpu_minus_oracle"] = max(tab["max_abs_log_duration_gpu_minus_oracle"], float(np.abs(gt_logw[u] - logw_t).max()))
    rel = np.concatenate(rel)
    tab["note"] = ("EMULATED ggml arithmetic, inferred from upstream ggerganov/ggml (ggml_vec_gelu_f32 with GGML_GELU_FP16, ggml_compute_forward_soft_max_f32 "
                   "with table_exp_f16 and a double sum); the maxilevi/ggml fork the reference builds against is absent. Stage one of this mode runs in the exact "
                   "order of include/vits_exact_math.h on both sides: gpu_vs_oracle_durations_differ must be 0. *_changed: ids of the benchmark batch whose "
                   "duration differs from the table-free arithmetic")
    return {"ids": int(total), "within_1e-3_of_a_ceil_boundary": int((rel < 1e-3).sum()), "within_1e-2_of_a_ceil_boundary": int((rel < 1e-2).sum()),
            "within_1e-4_of_a_ceil_boundary": int((rel < 1e-4).sum()), "smallest_relative_margin": float(rel.min()),
            "gpu_durations_equal_to_oracle": int(equal), "max_abs_log_duration_gpu_minus_oracle": maxdev,
            "latents_outside_the_spline_interval_q6": int(outside), "emulated_ggml_tables_q8": tab,
            "note": "relative distance of w = exp(log_duration) to the nearest integer, oracle stage one on every id of the benchmark batch; an implementation whose "
                    "log-durations differ from the oracle's by eps can change at most the ids counted within eps. Q6 (masked get / set misalignment outside the "
                    "spline interval) is reproduced literally in reference mode and is not exercised by this batch when the count is 0; Q8 (ggml's fp16 GELU / "
                    "soft-max tables) is measured by emulation, see emulated_ggml_tables_q8", "wall_s": time.perf_counter() - t0}


def sub_results(pkg, torch, base_model, base_bytes, mode, steps_scale=1.0):
    """The other configurations of BASELINE.json on the driver's clock: compact, library-default configuration (no
    per-kernel events), few steps each. c2 = batch 1 x 128 ids fp32 (ms per utterance); c3 in f16 / bf16 arithmetic; c5 = two
    resident bf16-stored models x 8 x 1024 ids in fp32 and in bf16 arithmetic. Each with the whole-path fraction of its roofs."""
    out = {}
    noise_base = 4321

    def run(models_ids, steps, warmup=2, pipelined=False):
        """models_ids: [(model, ids, seed, [out buffers], cap)]. pipelined: every model handle keeps two batches in flight
        (vits_model_submit_batch / vits_model_wait: stage one of batch i + 1 under the vocoder of batch i, bit-identical PCM); the timed
        region takes K batches in and K results out, pipeline fill and drain included."""
        def one(k):
            tot, frs = 0, []
            for m, ids, seed, bufs, cap in models_ids:
                _, lengths, frames = m.process_batch(ids, mode=mode, noise_kind=pkg.NOISE_COUNTER, noise_seed=seed, out_device=bufs[k % len(bufs)].data_ptr(),
                                                     out_device_stride=cap, skip_host_copy=True, keep_pcm=False)
                tot += int(lengths.sum())
                frs.append(frames)
            return tot, np.concatenate(frs)

        def submit(k):
            for m, ids, seed, bufs, cap in models_ids:
                m.submit_batch(ids, mode=mode, noise_seed=seed, out_device=bufs[k % len(bufs)].data_ptr(), out_device_stride=cap, skip_host_copy=True)

        def wait():
            tot, frs = 0, []
            for m, *_ in models_ids:
                _, lengths, frames = m.wait(keep_pcm=False)
                tot += int(lengths.sum())
                frs.append(frames)
            return tot, np.concatenate(frs)

        for k in range(warmup):
            one(k)
        if pipelined:
            # warm both pipeline slots (front-end stream, second stage-one arena, pinned staging are created on first use)
            submit(0)
            submit(1)
            wait()
            wait()
        torch.cuda.synchronize()
        # (no cyclic-GC pause inside a timed region: a generation-2 collection over the records of the main run — 40 ms with --steps 20 —
        # once landed in the 16 serial f16 steps and read as 16.4 instead of 14.0 ms per step)
        gc.collect()
        gc.disable()
        t = time.perf_counter()
        samples = 0
        if pipelined:
            submit(0)
            for k in range(1, steps):
                submit(k)
                n, frames = wait()
                samples += n
            n, frames = wait()
            samples += n
        else:
            for k in range(steps):
                n, frames = one(k)
                samples += n
        torch.cuda.synchronize()
        e = time.perf_counter() - t
        gc.enable()
        return e, samples, frames

    def measured_bytes(tag):
        """whole-step HBM bytes of a PMC artefact (profiles/*_pmc_traffic.json) collected for THIS build on the workload `tag`, or None"""
        want, best = pkg.source_sha16(), None
        for path in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")):
            try:
                with open(path) as fh:
                    d = json.load(fh)
            except Exception:
                continue
            if d.get("source_sha16") == want and d.get("workload_tag") == tag and d.get("whole_step") and (best is None or os.path.getmtime(path) > os.path.getmtime(best[0])):
                best = (path, d["whole_step"])
        return best

    def entry(e, samples, frames, steps, T, arith, utterances, tag=None):
        fl = sum(algorithmic_flops(T, int(f)) for f in frames) * steps
        sps = samples / e
        d = {"value": sps, "unit": "samples/s", "ms_per_step": 1000.0 * e / steps, "utterances_per_step": utterances, "ids_per_utterance": T, "steps": steps,
             "algorithmic_tflops": fl / e / 1e12}
        if arith == "f32":
            d.update({"binding_roof": "mfma_f32", "frac_of_binding_roof": fl / e / 1e12 / PEAK_F32_TFLOPS})
        elif arith == "f32split":
            # VITS_ARITH_F32_SPLIT: the resblock convs of the wide stages run as five bf16 MFMAs per 16 products (peak 2500 / 5 = 500 TFLOP/s-equivalent),
            # the rest of the path on the fp32 MFMA (157.3): the roof of the whole path is the FLOP-weighted harmonic blend of the two
            d.update({"binding_roof": "mfma_bf16_split_blend", "frac_of_binding_roof": fl / e / 1e12 / SPLIT_BLEND_PEAK_TFLOPS, "blend_peak_tflops": SPLIT_BLEND_PEAK_TFLOPS,
                      "dtype": "fp32-accurate results: bf16 x (2 weight, 3 activation) operand split on the bf16 matrix cores, fp32 accumulate (wide-stage resblock convs); rest exact fp32"})
        else:
            # HBM side: MEASURED bytes (PMC FETCH_SIZE x 2 + WRITE_SIZE over every launch of a step, tools/pmc_traffic.py) when an artefact of
            # this build and workload exists; otherwise the layer-granular MODEL of SURVEY 8(d), labelled as such — the fused kernels move
            # ~0.7 x the model's bytes, so the model figure overstates the HBM rate
            mf = fl / e / 1e12 / PEAK_MFMA16_TFLOPS
            art = measured_bytes(tag) if tag else None
            if art:
                per_sample = art[1]["hbm_bytes_per_step"] / max(art[1].get("samples_per_step") or (samples / steps), 1)
                hbm = per_sample * sps / 1e9 / PEAK_HBM_GBS
                d.update({"frac_hbm_measured": hbm, "hbm_bytes_per_sample_measured": per_sample, "hbm_source": os.path.relpath(art[0], ROOT)})
                d.update({"binding_roof": "hbm" if hbm >= mf else "mfma_16bit", "frac_of_binding_roof": max(hbm, mf), "frac_mfma16": mf})
            else:
                # no PMC artefact of this build for this workload: the HBM side is UNMEASURED and stays null (a byte model is not quoted as a
                # roof: the fused kernels move ~0.7 x the layer-granular model's bytes)
                d.update({"frac_hbm_measured": None, "hbm_source": None, "binding_roof": None, "frac_of_binding_roof": None, "frac_mfma16": mf})
        return d

    def buf_for(B, T, n=1):
        cap = 256 * 8 * T + 294
        return [torch.empty((B, cap), dtype=torch.float32, device="cuda") for _ in range(n)], cap

    PIPE_NOTE = ("ONE model handle, two batches in flight (vits_model_submit_batch / vits_model_wait): stage one of batch i + 1 on the handle's front-end "
                 "stream under the vocoder of batch i; PCM bit-identical to vits_model_process_batch (GPU test); K batches in and K results out inside the "
                 "timed region, fill and drain included")

    def both(name, models_ids, n, T, arith, utterances, warmup=2, tag=None):
        """serial calls (vits_model_process_batch, one batch in flight) and the pipelined schedule on the same handle(s); `value` is the
        pipelined one, the serial figure stays beside it"""
        e, s_, fr = run(models_ids, n, warmup=warmup)
        serial = entry(e, s_, fr, n, T, arith, utterances, tag)
        e, s_, fr = run(models_ids, n, warmup=1, pipelined=True)
        d = entry(e, s_, fr, n, T, arith, utterances, tag)
        d["schedule"] = PIPE_NOTE
        d["serial_calls"] = {k: serial[k] for k in ("value", "ms_per_step", "algorithmic_tflops", "frac_of_binding_roof")}
        out[name] = d

    # c2: batch 1, fp32
    ids1 = pkg.synth_ids(1, 128)
    b, cap = buf_for(1, 128)
    n = max(5, int(40 * steps_scale))
    e, s_, fr = run([(base_model, ids1, noise_base, b, cap)], n, warmup=5)
    out["c2_f32"] = entry(e, s_, fr, n, 128, "f32", 1)
    out["c2_f32"]["ms_per_utterance"] = 1000.0 * e / n
    out["c2_f32"]["schedule"] = "serial calls (latency figure: one utterance in, its PCM out)"

    def reference_api_ms(reps):
        """the reference's OWN entry point (vits_model_process_ids == vits_model_process behind the tokenizer: libstdc++ noise stream drawn on the host, fp32 PCM
        returned in host memory, /root/reference/src/vits.cpp:1225-1232), ms per utterance"""
        for _ in range(3):
            base_model.process_ids(ids1[0])
        gc.collect()
        gc.disable()
        t = time.perf_counter()
        for _ in range(reps):
            base_model.process_ids(ids1[0])
        e = time.perf_counter() - t
        gc.enable()
        return 1000.0 * e / reps

    out["c2_f32"]["reference_api_ms"] = reference_api_ms(n)
    b2, _ = buf_for(1, 128, 2)
    e, s_, fr = run([(base_model, ids1, noise_base, b2, cap)], n, warmup=1, pipelined=True)
    out["c2_f32"]["pipelined_throughput"] = {"value": s_ / e, "ms_per_utterance": 1000.0 * e / n,
                                             "note": "batch-1 utterances through the single-handle pipeline (two in flight): throughput, not latency"}
    # c2 in the reference's own conv arithmetic (fp16 operands, default scope): the latency of one utterance, serial calls
    base_model.set_arith(pkg.ARITH_F16)
    try:
        e, s_, fr = run([(base_model, ids1, noise_base, b, cap)], n, warmup=5)
        out["c2_f16"] = entry(e, s_, fr, n, 128, "f16", 1, tag="c2|b1|f16")
        out["c2_f16"]["ms_per_utterance"] = 1000.0 * e / n
        out["c2_f16"]["schedule"] = "serial calls (latency figure: one utterance in, its PCM out)"
        out["c2_f16"]["reference_api_ms"] = reference_api_ms(n)
    finally:
        base_model.set_arith(pkg.ARITH_F32)
    # c3 in fp32 (pipelined beside the headline's serial figure) and in the 16-bit arithmetic modes (default scope: stage one exact,
    # durations identical to the fp32 run's)
    ids64 = pkg.synth_ids(64, 128)
    b, cap = buf_for(64, 128, 2)
    n = max(3, int(6 * steps_scale))
    both("c3_f32", [(base_model, ids64, noise_base, b, cap)], n, 128, "f32", 64)
    for name, arith in (("f16", pkg.ARITH_F16), ("bf16", pkg.ARITH_BF16)):
        base_model.set_arith(arith)
        n = max(3, int(16 * steps_scale))
        both("c3_" + name, [(base_model, ids64, noise_base, b, cap)], n, 128, name, 64, tag="c3|b64|" + name)
    # c3 with fp32-accurate results from the bf16 matrix cores (VITS_ARITH_F32_SPLIT, csrc/conv_split.hip): opt-in, never the headline
    base_model.set_arith(pkg.ARITH_F32_SPLIT)
    n = max(3, int(6 * steps_scale))
    both("c3_f32split", [(base_model, ids64, noise_base, b, cap)], n, 128, "f32split", 64)
    base_model.set_arith(pkg.ARITH_F32)
    del b
    # c5: two resident bf16-stored models, 8 x 1024 ids each, calls interleaved
    specs = [(0x5EED, 1234), (0xBEEF, 91234)]
    ms = []
    for seed, ids_seed in specs:
        m = pkg.Model(pkg.synth_model_bytes(seed, pkg.SYNTH_FULL | pkg.SYNTH_BF16))
        m.set_mode(mode)
        bb, cap = buf_for(8, 1024, 2)
        ms.append((m, pkg.synth_ids(8, 1024, ids_seed=ids_seed), noise_base, bb, cap))
    try:
        for name, arith in (("f32", pkg.ARITH_F32), ("bf16", pkg.ARITH_BF16)):
            for m, *_ in ms:
                m.set_arith(arith)
            n = max(2, int((3 if name == "f32" else 6) * steps_scale))
            both("c5_" + name, ms, n, 1024, name, 16, warmup=1, tag="c5|b8|" + name)
    finally:
        for m, *_ in ms:
            m.close()
    return out



def serving_two_engines(pkg, torch, base_model, base_bytes, mode, steps=8):
    """NOT a BASELINE configuration — a serving data point: two resident engine instances of the same model on one GPU (the C API allows
    distinct handles to run concurrently), each fed whole batches of 64 x 128 ids by its own host thread. The device then overlaps the
    latency-bound stage one (text encoder + duration predictor, exact fp32) of one instance with the matrix-core-bound vocoder of the
    other; per-batch latency doubles. Reported beside, never instead of, the single-instance figures."""
    import threading
    out = {"note": "two engine instances (two model handles, two host threads), each processing whole 64 x 128-id batches; samples/s over both; "
                   "single-instance figures are `value` / sub_results"}
    other = pkg.Model(base_bytes)
    other.set_mode(mode)
    ids64 = pkg.synth_ids(64, 128)
    cap = 256 * 8 * 128 + 294
    bufs = [torch.empty((64, cap), dtype=torch.float32, device="cuda") for _ in range(2)]
    engines = [base_model, other]
    try:
        for name, arith in (("f32", pkg.ARITH_F32), ("f16", pkg.ARITH_F16), ("bf16", pkg.ARITH_BF16)):
            for m in engines:
                m.set_arith(arith)
            n = steps if name != "f32" else max(3, steps // 2)
            counts = [0, 0]

            def worker(i, reps):
                tot = 0
                for _ in range(reps):
                    _, lengths, _ = engines[i].process_batch(ids64, mode=mode, noise_kind=pkg.NOISE_COUNTER, noise_seed=4321, out_device=bufs[i].data_ptr(),
                                                             out_device_stride=cap, skip_host_copy=True, keep_pcm=False)
                    tot += int(lengths.sum())
                counts[i] = tot

            for i in range(2):
                worker(i, 1)  # warm-up (arena growth, first launches)
            torch.cuda.synchronize()
            t = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i, n)) for i in range(2)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            torch.cuda.synchronize()
            e = time.perf_counter() - t
            out["c3_" + name] = {"value": sum(counts) / e, "unit": "samples/s", "ms_per_batch": 1000.0 * e / (2 * n), "batches": 2 * n}
    finally:
        base_model.set_arith(pkg.ARITH_F32)
        other.close()
    return out


def find_profile_artifact(pkg, suffix, key, tag):
    """Newest profiles/*<suffix> that (a) records the source hash of the library sources of THIS run and (b) has an entry for the
    kernel `key` and (c) was collected on this workload (`tag` = workload|batch|arith: a kernel instantiation's bytes per launch and busy
    fraction depend on the shapes it ran on). The PMC passes are separate rocprofv3 runs over this same command; a file collected for
    another build or another workload is ignored — the fields then stay null."""
    want = pkg.source_sha16()
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)):
        try:
            with open(path) as fh:
                d = json.load(fh)
        except Exception:
            continue
        if d.get("source_sha16") == want and d.get("workload_tag") == tag and key in d.get("by_bench_key", {}) and (best is None or os.path.getmtime(path) > os.path.getmtime(best[0])):
            best = (path, d)
    return best


def default_schedule_roofline(pkg, model_bytes, mode, ids, noise_base, cap, out_buf, wall_ms_default, tag):
    """`roofline_default_schedule`: the roofline of what SHIPS — the library-default schedule (three streams, separate
    launches, no per-kernel events), which the instrumented headline region does not run (it serialises, and prefers the grouped launch).
      * per-kernel average durations and calls per step: a `rocprofv3 --kernel-trace --stats` of `bench.py --no-prof` reduced by
        tools/default_schedule.py (profiles/*_default_schedule.json; only an artefact of THIS build and workload is used);
      * algorithmic FLOP per launch: this run's own accounting on a second handle loaded with VITS_NO_RB_GROUP=1 (the same separate
        launches as the default schedule, serialised by the profiler), joined by kernel instantiation;
      * summed kernel time per step against the wall time of the default pass of this run = the overlap the three streams buy."""
    want, best = pkg.source_sha16(), None
    for path in glob.glob(os.path.join(ROOT, "profiles", "*_default_schedule.json")):
        try:
            with open(path) as fh:
                d = json.load(fh)
        except Exception:
            continue
        if d.get("source_sha16") == want and d.get("workload_tag") == tag and (best is None or os.path.getmtime(path) > os.path.getmtime(best[0])):
            best = (path, d)
    if best is None:
        return {"available": False, "note": "no profiles/*_default_schedule.json of this build (source_sha16 %s) and workload %s: run tools/jobs/profile.sh" % (want, tag)}
    art = best[1]
    os.environ["VITS_NO_RB_GROUP"] = "1"  # (knobs are read when a model is loaded)
    try:
        m2 = pkg.Model(model_bytes)
    finally:
        del os.environ["VITS_NO_RB_GROUP"]
    try:
        m2.set_mode(mode)
        kw = dict(mode=mode, noise_kind=pkg.NOISE_COUNTER, noise_seed=noise_base, out_device=out_buf.data_ptr(), out_device_stride=cap, skip_host_copy=True, keep_pcm=False)
        m2.process_batch(ids, **kw)
        m2.prof_reset()
        m2.prof_enable(True)
        m2.process_batch(ids, **kw)
        m2.prof_enable(False)
        rep = m2.prof_report()["kernels"]
    finally:
        m2.close()
    flop, calls = {}, {}
    for k in rep:
        parts = k["name"].split("|")
        key = "|".join(parts[1:5]) if len(parts) >= 5 else k["name"]
        flop[key] = flop.get(key, 0.0) + k["flop"]
        calls[key] = calls.get(key, 0) + k["calls"]
    rows, tot_flop, tot_ms, mismatched = [], 0.0, 0.0, []
    for key, e in art["by_bench_key"].items():
        if key not in flop or not flop[key]:
            continue
        if abs(calls[key] - e["calls_per_step"]) > 1e-6:
            mismatched.append(key)
            continue
        tf = flop[key] / (e["ms_per_step"] * 1e-3) / 1e12
        rows.append({"kernel": key, "rocprof_names": e["kernel_names"], "calls_per_step": e["calls_per_step"], "avg_launch_ms": e["avg_us"] / 1e3,
                     "ms_per_step": e["ms_per_step"], "algorithmic_tflops": tf, "frac_fp32_peak": tf / PEAK_F32_TFLOPS})
        tot_flop += flop[key]
        tot_ms += e["ms_per_step"]
    rows.sort(key=lambda r: -r["ms_per_step"])
    if not rows:
        return {"available": False, "note": "the artefact's kernels do not line up with this run's", "source": os.path.relpath(best[0], ROOT)}
    return {"available": True,
            "all_matrix_core_kernels": {"tflops": tot_flop / (tot_ms * 1e-3) / 1e12, "ms_per_step": tot_ms},
            "whole_schedule": {"algorithmic_tflops_of_the_listed_kernels_over_wall": tot_flop / (wall_ms_default * 1e-3) / 1e12 if wall_ms_default else None,
                               "frac_of_peak": tot_flop / (wall_ms_default * 1e-3) / 1e12 / PEAK_F32_TFLOPS if wall_ms_default else None},
            "summed_kernel_ms_per_step": art["summed_kernel_ms_per_step"], "wall_ms_per_step_this_run": wall_ms_default,
            "overlap_factor": art["summed_kernel_ms_per_step"] / wall_ms_default if wall_ms_default else None,
            "top_kernels": [{k: r[k] for k in ("kernel", "calls_per_step", "avg_launch_ms", "ms_per_step")} for r in rows[:16]],
            "kernels_with_other_call_counts": mismatched, "source": os.path.relpath(best[0], ROOT),
            "note": "library-default schedule (three streams, separate launches): durations from rocprofv3 --kernel-trace --stats of `bench.py --no-prof`, "
                    "FLOPs from this run's accounting of the same launches; overlap_factor > 1 = kernels of different streams ran concurrently. With three "
                    "streams a kernel shares the CUs with up to two others, so no per-kernel fraction is quoted here: `whole_schedule` (the listed kernels' "
                    "FLOPs over the wall time) is the rate of the schedule; the per-kernel roofline of exclusive launches is the instrumented `roofline` block"}


FINAL_LINE_LIMIT = 4096  # bytes; the driver keeps only a bounded tail of stdout (round 4: a 24 KB line came back as parsed: null)


# VITS_ARITH_F32_SPLIT: MACs per frame that run on the split kernels for the MMS-TTS architecture (ResBlocks C = 256 at rate 8, all taps 3 + 7 + 11;
# C = 128 at rate 64, all taps) against the path's 314.5 M (HiFiGAN 307.45 + flow 7.08); five bf16 MFMAs per 16
# products = 2500 / 5 TFLOP/s-equivalent for that share, the fp32 MFMA peak for the rest: harmonic blend
_SPLIT_MACS, _ALL_MACS = 6 * 256 * 256 * 21 * 8 + 6 * 128 * 128 * 21 * 64, 307453952 + 7077888
SPLIT_BLEND_PEAK_TFLOPS = _ALL_MACS / (_SPLIT_MACS / 500.0 + (_ALL_MACS - _SPLIT_MACS) / 157.3)
WORKLOAD_LABEL_MAX = 110  # the driver's record keeps 120 characters of config.workload (BENCH_r05 lost "mode=reference, conv arithmetic f32" at its end)


def workload_label(c5, B, T, mode, arith, pinned=0):
    """config.workload: at most WORKLOAD_LABEL_MAX characters, the two facts a reader needs — semantics mode and conv arithmetic — FIRST."""
    dur = "predicted-dur" if not pinned else "pinned-%d-frames/id" % pinned
    if c5:
        s = f"c5 2x(b{B}x{T}) {mode[:3]}-mode {arith} {dur} | two bf16-stored MMS-TTS-arch models, synthetic weights"
    else:
        s = f"c3 b{B}x{T} {mode[:3]}-mode {arith} {dur} | vits-english = MMS-TTS arch, synthetic weights"
    if B == 1 and not c5:
        s = "c2" + s[2:]
    assert len(s) <= WORKLOAD_LABEL_MAX, (len(s), s)
    return s


def _sig(x, n=6):
    """floats to n significant digits (the line is a record, not an archive: bench_detail.json keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (n, x))
    return x


def compact_line(res, detail_path=None):
    """The ONE line the driver parses: the contract's keys + roofline + cpu_baseline + one compact sub_results map, nothing else.
    Everything the run measured beyond that (top_kernels, roofline_default_schedule, notes, serving_two_engines, duration_boundary_margin)
    is in bench_detail.json. Pure function of the full result dict (tests/test_bench_host.py builds a worst case and checks the size)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "ranks_seen", "rccl_world_size", "devices_seen", "rtf", "algorithmic_tflops", "speedup_vs_cpu_baseline", "instrumented_ms_per_step")
    out = {k: _sig(res[k]) for k in keep if k in res}
    if isinstance(out.get("devices_seen"), list):
        out["devices_seen"] = out["devices_seen"][:16]
    out["metric"] = str(out.get("metric", ""))[:120]
    cfg = res.get("config", {})
    out["config"] = {k: (_sig(cfg[k]) if not isinstance(cfg[k], str) else cfg[k][:200]) for k in
                     ("workload", "batch_per_gpu", "ids_per_utterance", "samples_per_step", "parallelism", "source_sha16") if k in cfg}
    if isinstance(out["config"].get("workload"), str):
        out["config"]["workload"] = out["config"]["workload"][:WORKLOAD_LABEL_MAX]
    rf = res.get("roofline")
    if rf:
        r = {k: _sig(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_ms", "launches",
                                           "mfma_busy_frac", "share_of_gpu_time")}
        r["kernel"] = str(rf.get("kernel", ""))[:96]
        ws = rf.get("whole_step_traffic")
        r["whole_step_traffic_ratio"] = _sig(ws["ratio"]) if ws else None
        r["whole_step_frac_of_hbm_peak"] = _sig(ws["frac_of_hbm_peak"]) if ws else None
        for k in ("pmc_source",):
            if rf.get(k):
                r[k] = str(rf[k])[:80]
        out["roofline"] = r
    cb = res.get("cpu_baseline")
    if cb:
        c = {k: _sig(cb.get(k)) for k in ("value", "unit", "cores", "kind")}
        c["sample"] = str(cb.get("sample_short") or cb.get("sample", ""))[:160]
        c["threads"] = cb.get("threads")
        c["cpu_model"] = str(cb.get("cpu_model", ""))[:64]
        c["host_threads"] = cb.get("host_threads")
        c["at_reference_thread_rule_value"] = _sig((cb.get("at_reference_thread_rule") or {}).get("value"))
        c["at_reference_thread_rule_threads"] = (cb.get("at_reference_thread_rule") or {}).get("threads")
        c["one_thread_value"] = _sig((cb.get("one_thread") or {}).get("value"))
        out["cpu_baseline"] = c
    sub = res.get("sub_results")
    if sub:
        m = {}
        for name, d in sub.items():
            if not isinstance(d, dict) or "value" not in d:
                continue
            e = {"value": _sig(d["value"]), "ms_per_step": _sig(d.get("ms_per_step")), "frac_of_binding_roof": _sig(d.get("frac_of_binding_roof")),
                 "binding_roof": d.get("binding_roof"), "hbm_source": d.get("hbm_source")}
            if d.get("frac_mfma16") is not None:
                e["frac_mfma16"] = _sig(d["frac_mfma16"])
            if "serial_calls" in d:
                e["serial_ms_per_step"] = _sig(d["serial_calls"].get("ms_per_step"))
            if "reference_api_ms" in d:
                e["reference_api_ms"] = _sig(d["reference_api_ms"])
            m[name] = e
        out["sub_results"] = m
    dm = res.get("duration_boundary_margin")
    if dm:
        out["durations"] = {"ids": dm.get("ids"), "gpu_equal_to_oracle": dm.get("gpu_durations_equal_to_oracle"),
                            "ggml_tables_gpu_vs_oracle_differ": (dm.get("emulated_ggml_tables_q8") or {}).get("gpu_vs_oracle_durations_differ")}
    if detail_path:
        out["detail"] = detail_path
    line = json.dumps(out, separators=(",", ":"))
    # last resort (cannot happen with the field list above; the test builds the worst case): shed the optional blocks, never the contract's
    for k in ("durations", "sub_results"):
        if len(line) < FINAL_LINE_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) >= FINAL_LINE_LIMIT:
        raise RuntimeError("bench.py: final line is %d bytes" % len(line))
    return line


def write_detail(res):
    """the full result (every note, top_kernels, the default-schedule roofline ...) beside the script; gpurun_out/ as well when it exists, so that
    it travels back from the GPU box. Returns the repo-relative path that worked (or None)."""
    wrote = None
    for rel in ("bench_detail.json", os.path.join("gpurun_out", "bench_detail.json")):
        path = os.path.join(ROOT, rel)
        if not os.path.isdir(os.path.dirname(path)):
            continue
        try:
            with open(path, "w") as fh:
                json.dump(res, fh, indent=1)
            wrote = wrote or rel
        except OSError:
            pass
    return wrote


def launcher(args):
    """--gpus N without torchrun: start the N ranks as fresh child processes. Nothing here touches the GPU
    (torch.cuda.device_count() does not initialise it on this image), and no process that did is ever re-exec'd."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} device(s) visible; refusing to run fewer ranks\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # watch every rank: one that dies would leave the others waiting in the collective forever
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.kill()  # (our own children, by handle)
    out0 = procs[0].stdout.read() if procs[0].stdout else b""
    codes = [p.wait() for p in procs]
    if any(codes):
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        return 1
    line = [ln for ln in out0.decode().splitlines() if ln.startswith("{")]
    if not line:
        sys.stderr.write("bench.py: rank 0 printed no result\n")
        return 1
    print(line[-1])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c3", "c5"], default="c3",
                    help="c3 (BASELINE.json metric): one model, batch 64 x 128 ids. c5: two resident models with bf16-stored weights, "
                         "1024-id utterances, calls interleaved (BASELINE.json configs[4])")
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (per model); default 64 (c3) / 8 (c5)")
    ap.add_argument("--ids-per-utt", type=int, default=0, help="encoder input ids per utterance (after blank interspersing); default 128 (c3) / 1024 (c5)")
    ap.add_argument("--pinned", type=int, default=0, help=">0: pin every id to this many frames (SURVEY §8d run ii)")
    ap.add_argument("--mode", choices=["reference", "hf"], default="reference")
    ap.add_argument("--arith", choices=["f32", "bf16", "f16", "f32split"], default="f32",
                    help="conv operand precision: f32 (exact, default), or 16-bit MFMA operands with fp32 accumulation (vits_model_set_arith)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--no-extra-passes", action="store_true", help="skip the pinned-duration and host-PCM passes reported beside the headline")
    ap.add_argument("--no-sub-results", action="store_true", help="skip the compact sub-results (c2_f32, c2_f16, c3_f16, c3_bf16, c5_f32, c5_bf16) and the "
                    "duration-boundary report that the default single-GPU run prints beside the headline")
    ap.add_argument("--no-serving", action="store_true", help="skip the two-engine serving data point (`serving_two_engines`) of the default run")
    ap.add_argument("--pcm16", action="store_true", help="multi-GPU: convert to int16 on the device and gather that (half the bytes)")
    ap.add_argument("--balance", choices=["none", "frames"], default="none",
                    help="multi-GPU: 'frames' = every step first predicts frames (frames_only pre-pass on the own block), all-gathers "
                         "them, and re-shards the global batch so that the per-rank frame sums are equal (SURVEY §8e)")
    ap.add_argument("--chunk-frames", type=int, default=0,
                    help="run the vocoder in windows of this many frames (vits_process_opts.vocoder_chunk_frames); 0 = whole utterance")
    ap.add_argument("--single-pass", action="store_true",
                    help="instrumented (serialised) warmup + timed region only, no further passes: every launch of the run is then one "
                         "the HIP events timed, which is what a rocprofv3 --kernel-trace of this command is compared against")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus > 1 or os.environ.get("VITS_BENCH_LAUNCH") == "1"):  # (VITS_BENCH_LAUNCH=1: launcher path at N = 1, for tests)
        sys.exit(launcher(args))
    world = int(env_world or "1")
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}\n")
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank}")
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

    c5 = args.workload == "c5"
    B = args.batch or (8 if c5 else 64)
    T = args.ids_per_utt or (1024 if c5 else 128)
    mode = pkg.MODE_REFERENCE if args.mode == "reference" else pkg.MODE_HF
    # c3: "vits-english"; c5: "vits-spanish + english": two synthetic weight sets resident at once, conv weights stored as bf16
    model_specs = [(0x5EED, pkg.SYNTH_FULL | pkg.SYNTH_BF16, 1234), (0xBEEF, pkg.SYNTH_FULL | pkg.SYNTH_BF16, 91234)] if c5 else [(0x5EED, pkg.SYNTH_FULL, 1234)]
    arith = {"f32": pkg.ARITH_F32, "bf16": pkg.ARITH_BF16, "f16": pkg.ARITH_F16, "f32split": pkg.ARITH_F32_SPLIT}[args.arith]
    cap = 256 * 8 * T + 294  # PCM row capacity: up to 8 frames per id
    jobs = []
    for seed, arch, ids_seed in model_specs:
        mb = pkg.synth_model_bytes(seed, arch)
        m = pkg.Model(mb)
        m.set_mode(mode)
        if arith != pkg.ARITH_F32:
            m.set_arith(arith)
        ids_all = pkg.synth_ids(B * world, T, ids_seed=ids_seed)  # global batch; rank r owns [r*B, (r+1)*B) unless --balance
        # the PCM of step i is read by the exchange while step i + 1 is synthesised: three output buffers rotate (multi_gpu.PcmExchange)
        nbuf = 3 if dist_on else 1
        jobs.append({"model": m, "bytes": mb, "ids_all": ids_all, "ids": ids_all[rank * B:(rank + 1) * B], "offsets": None,
                     "outs": [torch.empty((B, cap), dtype=torch.float32, device="cuda") for _ in range(nbuf)],
                     "outs16": [torch.zeros((B, cap), dtype=torch.int16, device="cuda") for _ in range(nbuf)] if (dist_on and args.pcm16) else None,
                     "ex": None, "n": 0})
    if dist_on:
        for j in jobs:
            j["ex"] = mg.PcmExchange(B, cap, dtype=torch.int16 if args.pcm16 else torch.float32, device="cuda")
    noise_base = 4321  # global utterance u draws from the counter stream with seed 4321 + u, wherever it runs

    def step(host_pcm=False, pinned=0, collect=None):
        pinned = pinned or args.pinned
        tot_len, tot_frames = [], []
        for j in jobs:
            m, ids, offs = j["model"], j["ids"], j["offsets"]
            if dist_on and args.balance == "frames" and world > 1:
                # dispatcher pre-pass: predicted frames of the own block -> all ranks -> identical balanced assignment
                _, _, fr = m.process_batch(ids, mode=mode, noise_kind=pkg.NOISE_COUNTER, noise_seed=noise_base + rank * B, fixed_duration=pinned, frames_only=True)
                all_fr = mg.gather_frames(torch.from_numpy(fr).cuda()).cpu().numpy()
                mine = mg.balanced_shards(all_fr, world)[rank]
                ids, offs = j["ids_all"][mine], np.asarray(mine, np.int32)
            if offs is None:
                kw = {"noise_seed": noise_base + rank * B}
            else:
                kw = {"noise_seed": noise_base, "noise_seed_offsets": offs}
            out = j["outs"][j["n"] % len(j["outs"])]
            _, lengths, frames = m.process_batch(ids, mode=mode, noise_kind=pkg.NOISE_COUNTER, fixed_duration=pinned, out_device=out.data_ptr(),
                                                 out_device_stride=cap, skip_host_copy=not host_pcm, keep_pcm=False,
                                                 vocoder_chunk_frames=args.chunk_frames, **kw)
            if dist_on:
                # the path's only exchange: ragged all-gather of the PCM over RCCL/xGMI, pipelined — the lengths of this step are
                # queued now (fixed-size, no host round trip), the PCM of the PREVIOUS step is gathered while the next one is
                # synthesised; the last one is drained inside the timed region (see timed()). No meta exchange, no allocation.
                lens_d = torch.from_numpy(lengths).cuda()
                if args.pcm16:
                    o16 = j["outs16"][j["n"] % len(j["outs16"])]
                    pkg.pcm16_device(out.data_ptr(), out.stride(0), o16.data_ptr(), o16.stride(0), B, cap, lengths_ptr=lens_d.data_ptr(),
                                     stream=torch.cuda.current_stream().cuda_stream)
                    out = o16
                j["ex"].on_block = (lambda step, g, gl: collect.append((g.clone(), gl.clone()))) if collect is not None else None
                j["ex"].submit(out, lens_d)
            j["n"] += 1
            tot_len.append(lengths)
            tot_frames.append(frames)
        return np.concatenate(tot_len), np.concatenate(tot_frames)

    def fence():
        for j in jobs:
            if j["ex"] is not None:
                j["ex"].flush()  # drain the exchange pipeline: the last step's PCM gather belongs to the region it was submitted in
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, **kw):
        fence()
        gc.collect()
        gc.disable()  # (no cyclic-GC pause inside the timed region)
        t = time.perf_counter()
        samples = 0
        for _ in range(n):
            lengths, frames = step(**kw)
            samples += int(lengths.sum())
        fence()
        e = time.perf_counter() - t
        gc.enable()
        return e, samples, frames

    models = [j["model"] for j in jobs]
    if args.single_pass and not args.no_prof:
        for m in models:
            m.prof_enable(True)
    for _ in range(args.warmup):
        step()
    fence()
    # The timed region of the contract — EXACTLY K steps between barrier + synchronize — runs the library's DEFAULT schedule (no per-kernel events:
    # the three resblocks of a vocoder stage on concurrent streams). Rounds 1-5 timed the instrumented schedule here and paid 2.4 % of the headline
    # for their own instrumentation. The per-kernel HIP events of the roofline block come from a SECOND pass of the same K steps
    # right behind it (every rank runs it: the steps contain the PCM exchange), whose step time is reported as `instrumented_ms_per_step`.
    # --single-pass (the rocprofv3 runs of tools/jobs/profile.sh: every launch of the trace must be a serialised one) keeps ONE instrumented region.
    extra = {}
    if args.single_pass and not args.no_prof:
        for m in models:
            m.prof_reset()
        elapsed, total_samples, frames = timed(args.steps)
    else:
        elapsed, total_samples, frames = timed(args.steps)
        if not args.no_prof:
            for m in models:
                m.prof_reset()
                m.prof_enable(True)
            step()  # (the first serialised step re-lays the grouped launch out; untimed)
            for m in models:
                m.prof_reset()
            e, s_, _ = timed(args.steps)
            extra["instrumented"] = (e, s_)
    for m in models:
        m.prof_enable(False)
    # further passes of the same K steps, reported BESIDE the headline: (b) durations pinned to 2 frames per id (SURVEY §8d run ii: equal
    # lengths, L = 2T); (c) the PCM also copied to host memory (what the drop-in vits_model_process returns) — PCIe-inclusive, never `value`.
    if not args.single_pass and world == 1:
        if not args.no_extra_passes:
            step(pinned=2)
            e, s, fr = timed(args.steps, pinned=2)
            extra["pinned"] = (e, s, fr)
            step(host_pcm=True)
            e, s, _ = timed(args.steps, host_pcm=True)
            extra["host"] = (e, s)
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ts = torch.tensor([total_samples], dtype=torch.int64, device="cuda")
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_samples = int(ts.item())
    # self-check of an N-GPU run: the number of ranks that actually took part, counted THROUGH the communicator (a sum of ones over
    # RCCL), and the devices they ran on — so that a line that claims n_gpus = 8 cannot come from fewer processes
    ranks_seen, devices_seen = 1, [torch.cuda.current_device()]
    if dist_on:
        one = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
        dv = torch.full((dist.get_world_size(),), -1, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(dv, torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device="cuda"))
        devices_seen = [int(x) for x in dv.tolist()]

    res = None
    if rank == 0:
        sr = models[0].sampling_rate
        value = total_samples / elapsed
        flops_step = sum(algorithmic_flops(T, int(f)) for f in frames) * world
        dur_txt = "predicted" if not args.pinned else "pinned %d frames/id" % args.pinned
        workload = workload_label(c5, B, T, args.mode, args.arith, args.pinned)
        if c5:
            workload_long = (f"vits-spanish + english stand-ins: TWO resident MMS-TTS-architecture models (synthetic weights, seeds 0x5EED / 0xBEEF, conv weights "
                             f"stored as bf16), {B} utterances x {T} ids per model per step, calls interleaved, {dur_txt} durations, mode={args.mode}, conv arithmetic {args.arith}")
        else:
            workload_long = (f"vits-english (MMS-TTS architecture, synthetic weights), batch={B} per GPU, {T} ids per utterance, {dur_txt} durations, "
                             f"mode={args.mode}, conv arithmetic {args.arith}")
        res = {
            "metric": "audio samples/sec end-to-end (ids -> fp32 PCM in HBM), vits-english architecture, batch=64 x 128 ids" if not c5 else
                      "audio samples/sec end-to-end (ids -> fp32 PCM in HBM), two resident bf16-stored models, 1024-id utterances",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.arith if args.arith != "f32split" else "f32 (bf16 operand split on the matrix cores, fp32 accumulate)", "data": "synthetic", "ranks_seen": ranks_seen, "rccl_world_size": dist.get_world_size() if dist_on else 1,
            "devices_seen": devices_seen,
            "config": {"workload": workload, "workload_long": workload_long, "batch_per_gpu": B * len(jobs), "ids_per_utterance": T, "frames_per_utterance_mean": float(np.mean(frames)),
                       "samples_per_step": total_samples // args.steps, "sampling_rate": sr, "parallelism": f"utterance-sharded x{world}",
                       "shard_balance": args.balance, "pcm_destination": "device (HBM)", "vocoder_chunk_frames": args.chunk_frames,
                       "source_sha16": pkg.source_sha16()},
            "rtf": elapsed / (total_samples / float(sr)), "rtf_22050": elapsed / (total_samples / 22050.0),
            "algorithmic_tflops": flops_step * args.steps / elapsed / 1e12,
        }
        # whole-path fraction of the matrix-core peak of the arithmetic in use (fp32: 157.3 TFLOP/s; 16-bit operands: 2.5 PFLOP/s dense)
        res["frac_fp32_peak_whole_path" if args.arith == "f32" else ("frac_split_blend_peak_whole_path" if args.arith == "f32split" else "frac_mfma16_peak_whole_path")] = \
            flops_step * args.steps / elapsed / 1e12 / ((PEAK_F32_TFLOPS if args.arith == "f32" else SPLIT_BLEND_PEAK_TFLOPS if args.arith == "f32split" else 2500.0) * world)
        res["schedule"] = ("instrumented (single pass under a profiler): per-kernel HIP events, resblocks of a stage serialised" if args.single_pass and not args.no_prof
                           else "library default: no per-kernel events, resblocks of a stage on 3 concurrent streams")
        if "instrumented" in extra:
            e, s = extra["instrumented"]
            res["instrumented_value"] = s / e
            res["instrumented_ms_per_step"] = 1000.0 * e / args.steps
            res["instrumented_note"] = ("second pass of the same K steps with a pair of HIP events per kernel launch (the roofline block's durations) and the resblocks of a "
                                        "stage serialised: what rounds 1-5 reported as the headline")
        if "pinned" in extra:
            e, s, fr = extra["pinned"]
            fl = sum(algorithmic_flops(T, int(f)) for f in fr)
            res["pinned_durations"] = {"value": s / e, "ms_per_step": 1000.0 * e / args.steps, "samples_per_step": s // args.steps, "frames_per_id": 2,
                                       "algorithmic_tflops": fl * args.steps / e / 1e12,
                                       "frac_peak_whole_path": fl * args.steps / e / 1e12 / (PEAK_F32_TFLOPS if args.arith == "f32" else 2500.0),
                                       "note": "SURVEY 8d run (ii): every id lasts 2 frames (equal lengths), library default configuration"}
        if "host" in extra:
            e, s = extra["host"]
            res["host_pcm"] = {"value": s / e, "ms_per_step": 1000.0 * e / args.steps,
                               "note": "PCM also copied to pageable host memory (what the drop-in vits_model_process returns): PCIe-inclusive, never the headline"}
        if not args.no_prof:
            rep = []
            for m in models:
                rep += m.prof_report()["kernels"]
            # group by kernel instantiation (taps, dilation, tile, epilogue) == one rocprofv3 kernel name
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
            # dominant kernel: among the matrix-core kernels (the ones with algorithmic FLOP / byte accounting; with 1024-id inputs the
            # VALU attention kernel can be the longest single entry of a 16-bit step, and it has no MFMA / HBM roof to be priced against).
            # fp32: one template instantiation == one rocprofv3 kernel name. 16-bit modes: every fused pair / layer is its own
            # instantiation (none above 10 % of the step), so the instantiations are first aggregated by kernel FAMILY (the tile letter:
            # F = rbpair16, T = conv16, W = wavenet16, ...) and the roofline describes the dominant family — per launch averages over
            # its instantiations, traffic from the PMC file summed the same way.
            mfma_keys = [kk for kk in groups if kk.startswith("k") and groups[kk]["flop"] > 0]
            family_of = lambda kk: (kk.split("|")[2][0] if len(kk.split("|")) >= 3 else "?")
            FAMILY_NAMES = {"F": "rbpair16_kernel (fused ResBlock conv pair)", "T": "conv16_kernel", "W": "wavenet16_kernel", "f": "rbpair32_kernel",
                            "t": "conv_mfma_kernel", "w": "wavenet32_kernel", "G": "conv_group_kernel", "B": "rbblock16_kernel (whole ResBlock)", "b": "rbblock32_kernel (whole 3-tap ResBlock, fp32)",
                            "C": "flow_couple16_kernel (whole coupling layer)"}
            dom_members = None
            if args.arith not in ("f32", "f32split") and mfma_keys:
                fam = {}
                for kk in mfma_keys:
                    fam.setdefault(family_of(kk), []).append(kk)
                dom_fam = max(fam, key=lambda f: sum(groups[kk]["ms"] for kk in fam[f]))
                dom_members = fam[dom_fam]
                dom = {"calls": sum(groups[kk]["calls"] for kk in dom_members), "ms": sum(groups[kk]["ms"] for kk in dom_members),
                       "flop": sum(groups[kk]["flop"] for kk in dom_members), "bytes": sum(groups[kk]["bytes"] for kk in dom_members)}
                dom_key = "family " + dom_fam + ": " + FAMILY_NAMES.get(dom_fam, dom_fam) + " (" + ", ".join(sorted(dom_members)) + ")"
            else:
                dom_key = max(mfma_keys or groups, key=lambda kk: groups[kk]["ms"])
                dom = groups[dom_key]
                dom_members = [dom_key]
            avg_ms = dom["ms"] / dom["calls"]
            achieved = dom["flop"] / dom["calls"] / (avg_ms * 1e-3) / 1e12
            all_ms = sum(g["ms"] for g in groups.values())
            peak = PEAK_F32_TFLOPS if args.arith == "f32" else PEAK_MFMA16_TFLOPS  # dense MFMA peak of the operand type (MI355X_MICROARCH.md)
            if args.arith == "f32split":  # the split kernels (tile letter S): five bf16 MFMAs per 16 products; every other kernel of this mode is the fp32 path
                peak = PEAK_MFMA16_TFLOPS / 5.0 if family_of(dom_members[0]) == "S" else PEAK_F32_TFLOPS
            alg_gbs = dom["bytes"] / dom["calls"] / (avg_ms * 1e-3) / 1e9
            mfma_frac, hbm_frac = achieved / peak, alg_gbs / PEAK_HBM_GBS
            if args.arith in ("f32", "f32split") or mfma_frac >= hbm_frac:
                res["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": mfma_frac}
            else:  # 16-bit operands: 16x the MFMA rate, the same bytes -> the conv is bound by HBM, not by the matrix cores
                res["roofline"] = {"bound": "hbm", "achieved": alg_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_frac}
            if args.arith not in ("f32", "f32split"):
                # context for the 16-bit fractions: a register-only loop of v_mfma_f32_32x32x16_{f16,bf16} sustains 1.67 (f16) /
                # 1.80 (bf16) PFLOP/s at 1.63 / 1.78 GHz on operands with random signs and exponents spread over 2^-7..2^0, 1.96 PFLOP/s on
                # same-sign same-exponent operands (tools/mfma16_peak.hip) - the power budget, not the issue rate, sets the ceiling
                res["roofline"]["sustained_mfma16_tflops_measured"] = 1800.0 if args.arith == "bf16" else 1670.0
                res["roofline"]["sustained_note"] = ("tools/mfma16_peak.hip, operands with random signs and spread exponents, no memory traffic; "
                                                     "`peak` stays the guide's dense figure")
            res["roofline"].update({"traffic": None, "kernel": dom_key, "avg_launch_ms": avg_ms, "launches": dom["calls"],
                                    "share_of_gpu_time": dom["ms"] / all_ms,
                                    "flop_accounting": "algorithmic: 2 * rows * c_in * taps * (sum over utterances of the real output length), not the padded grid",
                                    "algorithmic_tflops": achieved, "mfma_frac_if_algorithmic": mfma_frac,
                                    "algorithmic_gbytes_per_launch": dom["bytes"] / dom["calls"] / 1e9,
                                    "hbm_frac_if_algorithmic": hbm_frac})
            # HBM bytes per launch of that kernel (family: launch-weighted mean over its instantiations) and MFMA-busy fraction from the
            # PMC passes (separate rocprofv3 --pmc runs over this same command, reduced by tools/pmc_traffic.py / tools/pmc_mfma.py):
            # only a file collected for THIS build (same source hash) and THIS workload is used, otherwise the fields stay null
            pmc_tag = "%s|b%d|%s" % (args.workload, B, args.arith)
            art = find_profile_artifact(pkg, "_pmc_traffic.json", dom_members[0], pmc_tag)
            if art and all(kk in art[1].get("by_bench_key", {}) for kk in dom_members):
                ents = art[1]["by_bench_key"]
                tot = sum(ents[kk]["hbm_bytes_per_launch"] * groups[kk]["calls"] for kk in dom_members)
                res["roofline"]["traffic"] = tot / dom["calls"]
                res["roofline"]["traffic_unit"] = "bytes per launch (" + art[1].get("calibration", "PMC") + "; " + os.path.relpath(art[0], ROOT) + ")"
                res["roofline"]["traffic_over_algorithmic"] = tot / dom["bytes"]
                # whole step: every kernel of the trace (PMC bytes summed over all launches / steps in the trace) against the
                # layer-granular algorithmic model of SURVEY 8(d) — the figure that matters where the path is HBM-bound
                ws = art[1].get("whole_step")
                if ws:
                    sps = total_samples / args.steps / world
                    alg = (ALG_BYTES_PER_SAMPLE_F32 if args.arith in ("f32", "f32split") else ALG_BYTES_PER_SAMPLE_16) * sps
                    res["roofline"]["whole_step_traffic"] = {
                        "hbm_bytes_per_step": ws["hbm_bytes_per_step"], "fetch_bytes_per_step": ws["fetch_bytes_per_step"], "write_bytes_per_step": ws["write_bytes_per_step"],
                        "algorithmic_bytes_per_step": alg, "algorithmic_model": "%.1f KB per output sample (SURVEY 8d layer-granular activation model%s) x %d samples" % (
                            (ALG_BYTES_PER_SAMPLE_F32 if args.arith in ("f32", "f32split") else ALG_BYTES_PER_SAMPLE_16) / 1e3, "" if args.arith in ("f32", "f32split") else ", 16-bit conv inputs: x 16/24", sps),
                        "ratio": ws["hbm_bytes_per_step"] / alg, "samples_per_step_in_trace": ws.get("samples_per_step"),
                        "hbm_gbs_at_this_step_time": ws["hbm_bytes_per_step"] / (elapsed / args.steps) / 1e9,
                        "frac_of_hbm_peak": ws["hbm_bytes_per_step"] / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS, "source": os.path.relpath(art[0], ROOT)}
            art = find_profile_artifact(pkg, "_pmc_mfma.json", dom_members[0], pmc_tag)
            if art and all(kk in art[1].get("by_bench_key", {}) for kk in dom_members):
                ents = art[1]["by_bench_key"]
                wsum = lambda f: (sum(ents[kk].get(f) * groups[kk]["ms"] for kk in dom_members) / dom["ms"]) if all(ents[kk].get(f) is not None for kk in dom_members) else None
                res["roofline"]["mfma_busy_frac"] = wsum("mfma_busy_frac")
                res["roofline"]["lds_bank_conflict_frac"] = wsum("lds_bank_conflict_frac")
                # the shader clock the dominant launches ran at (GRBM_GUI_ACTIVE / 8 XCDs / duration, tools/pmc_mfma.py): 2.36-2.42 GHz for the fp32
                # kernels, 1.25-1.8 GHz for the MFMA-dense 16-bit ones (power budget) — the nominal peaks assume 2.4 GHz
                res["roofline"]["clock_ghz_estimate"] = wsum("clock_ghz_estimate")
                res["roofline"]["mfma_busy_frac_of_elapsed_clocks"] = wsum("mfma_busy_frac_of_elapsed_clocks")
                res["roofline"]["pmc_source"] = os.path.relpath(art[0], ROOT)
            conv_ms = sum(g["ms"] for kk, g in groups.items() if kk.startswith("k"))
            conv_flop = sum(g["flop"] for kk, g in groups.items() if kk.startswith("k"))
            res["kernel_time_ms_per_step"] = all_ms / args.steps
            res["all_conv_kernels"] = {"tflops": conv_flop / (conv_ms * 1e-3) / 1e12, "frac_of_peak": conv_flop / (conv_ms * 1e-3) / 1e12 / peak,
                                       "share_of_gpu_time": conv_ms / all_ms}
            top = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:48]
            res["top_kernels"] = [{"kernel": kk, "ms_per_step": g["ms"] / args.steps, "calls_per_step": g["calls"] / args.steps,
                                   "tflops": (g["flop"] / (g["ms"] * 1e-3) / 1e12) if g["flop"] else None,
                                   "algorithmic_gbs": (g["bytes"] / (g["ms"] * 1e-3) / 1e9) if g["bytes"] else None} for kk, g in top]
        default_run = world == 1 and not c5 and args.arith == "f32" and not args.batch and not args.ids_per_utt and not args.pinned and not args.single_pass and \
            not args.chunk_frames and not dist_on
        if default_run and "instrumented" in extra:
            res["roofline_default_schedule"] = default_schedule_roofline(pkg, jobs[0]["bytes"], mode, jobs[0]["ids"], noise_base, cap, jobs[0]["outs"][0],
                                                                         res.get("ms_per_step"), "c3|b%d|f32" % B)
        if default_run and not args.no_sub_results:
            t0 = time.perf_counter()
            res["sub_results"] = sub_results(pkg, torch, models[0], jobs[0]["bytes"], mode)
            if not args.no_serving:
                res["serving_two_engines"] = serving_two_engines(pkg, torch, models[0], jobs[0]["bytes"], mode)
            res["sub_results"]["wall_s"] = time.perf_counter() - t0
            res["sub_results"]["note"] = ("library default configuration (no per-kernel events), ids -> fp32 PCM in HBM, reference mode, predicted durations; 16-bit modes: "
                                          "VITS_ARITH_SCOPE_FLOW_VOCODER (stage one exact fp32, durations identical to the fp32 run)")
            if not args.no_cpu_baseline:
                res["duration_boundary_margin"] = duration_boundary_margin(pkg, models[0], jobs[0]["bytes"], jobs[0]["ids"], noise_base, mode, args.mode)
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = 24 if not c5 else 1
            cb = cpu_baseline([(j["bytes"], j["ids"][:n_cpu], noise_base) for j in jobs], args.mode, with_one_thread=not c5)
            res["cpu_baseline"] = cb
            res["speedup_vs_cpu_baseline"] = value / cb["value"]
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    for m in models:
        m.close()
    if rank == 0:
        # the JSON line goes out LAST: RCCL writes a version banner to the C stdout, which would otherwise follow it
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(compact_line(res, write_detail(res)), flush=True)


if __name__ == "__main__":
    main()
