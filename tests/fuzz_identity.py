"""Randomised identity / parity run over the C ABI (test helper, GPU; run by tests/test_gpu_pipeline.py with a small budget and by hand with a large one): for random batches, lengths, modes, arithmetic modes and window
sizes checks the properties the library promises, bit for bit —
  * a row of a ragged batch == the same utterance alone (noise stream keyed by seed + row),
  * windowed vocoder == whole vocoder, pipelined submit / wait == process_batch, a second handle == the first,
  * vits_model_process_batch split in two pipelined parts inside the call (the main handle is loaded with VITS_SPLIT_MIN_BATCH=2 and an
    uneven VITS_SPLIT_FIRST_PCT) == the unsplit call of a handle loaded with VITS_SPLIT_MIN_BATCH=0 — whose whole-resblock kernels also walk
    segments of three tiles (VITS_RBB_STREAM_*: random lengths put the sequence end anywhere in a segment),
  * 16-bit modes (default scope) and the split arithmetic (VITS_ARITH_F32_SPLIT): durations and frame counts == the fp32 run,
and, for the small architecture, float parity with the oracle in the same arithmetic (waveform <= tol x RMS, durations exact).
usage: python tests/fuzz_identity.py [--trials N] [--seed S] [--no-oracle]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package, rel_err

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--no-oracle", action="store_true")
args = ap.parse_args()
pkg = load_package()
oracle = None
if not args.no_oracle:
    import oracle_lib as oracle
    oracle.lib()
rng = np.random.default_rng(args.seed)
ARITH = {"f32": pkg.ARITH_F32, "f16": pkg.ARITH_F16, "bf16": pkg.ARITH_BF16, "f32split": pkg.ARITH_F32_SPLIT}
ORACLE_ARITH = {"f32": 0, "bf16": 1, "f16": 2, "f32split": 0}  # == oracle_lib.ARITH_*; the split arithmetic (round 6) is held to the fp32 oracle
models, oracles = {}, {}


def model(arch, arith, ref=False):
    """ONE handle per architecture that switches its arithmetic from trial to trial (vits_model_set_arith back and forth), and a
    second one that stays in fp32 (the reference of the 16-bit duration check)."""
    key = (arch, ref)
    if key not in models:
        # (knobs are read when a model is loaded) main handle: every batch of two or more is split 37 : 63; ref / "unsplit" handles: never
        os.environ["VITS_SPLIT_MIN_BATCH"] = "0" if ref else "2"
        os.environ["VITS_SPLIT_FIRST_PCT"] = "37"
        if ref == "unsplit":  # ... and its whole-resblock kernels (16-bit modes) walk segments of three tiles from two tiles up, the main handle's one tile per block at these sizes
            os.environ["VITS_RBB_STREAM_MIN_BLOCKS"], os.environ["VITS_RBB_STREAM_TILES"] = "1", "3"
            os.environ["VITS_NO_LAT16H"] = "1"  # ... and the wide stages' resblock convs never take conv16_lat_kernel (the main handle's do on small grids)
            os.environ["VITS_NO_RBB_GROUP3"] = "1"  # ... and the narrow 16-bit stages' whole-resblock kernels are three launches on three streams (the main handle: one grouped launch on small grids)
            os.environ["VITS_NO_RB_SUM3_F32"] = "1"  # ... and the fp32 vocoder's resblocks always chain through the shared sum (the main handle's run side by side on small grids)
        try:
            models[key] = [pkg.Model(pkg.synth_model_bytes(0x5EED, arch)), "f32", False]
        finally:
            del os.environ["VITS_SPLIT_MIN_BATCH"], os.environ["VITS_SPLIT_FIRST_PCT"]
            os.environ.pop("VITS_RBB_STREAM_MIN_BLOCKS", None), os.environ.pop("VITS_RBB_STREAM_TILES", None), os.environ.pop("VITS_NO_LAT16H", None), os.environ.pop("VITS_NO_RB_SUM3_F32", None), os.environ.pop("VITS_NO_RBB_GROUP3", None)
    e = models[key]
    if e[1] != arith:
        e[0].set_arith(ARITH[arith])
        e[1] = arith
    return e[0]


def tables(arch, on):
    for ref in (False, True, "unsplit"):
        e = models.get((arch, ref))
        if e and e[2] != on:
            e[0].set_ggml_tables(on)
            e[2] = on


def same(a, b, what, ctx):
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (what, "lengths / frames", ctx, a[1], b[1])
    for r, (x, y) in enumerate(zip(a[0], b[0])):
        if not np.array_equal(x, y):
            d = np.flatnonzero(x != y)
            raise AssertionError((what, "pcm row %d: %d of %d samples differ, first at %d" % (r, d.size, x.size, d[0]), ctx))


t0 = time.time()
counts = {"trials": 0, "single": 0, "windowed": 0, "pipelined": 0, "split": 0, "dur16": 0, "oracle": 0}
worst = 0.0
for trial in range(args.trials):
    arch = pkg.SYNTH_TINY if rng.random() < 0.4 else pkg.SYNTH_FULL
    arith = ["f32", "f16", "bf16", "f32split"][int(rng.integers(4))]
    mode = int(rng.integers(2))
    B = int(rng.integers(1, 9))
    L = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 40, 64, 130, 300, 700], p=[.05, .05, .05, .1, .15, .2, .15, .1, .08, .04, .02, .01]))
    lens = rng.integers(1, L + 1, size=B).astype(np.int32)
    if rng.random() < 0.3:
        lens[:] = L
    lens[int(rng.integers(B))] = L
    ids = pkg.synth_ids(B, L, ids_seed=int(rng.integers(1 << 30)))
    seed = int(rng.integers(1 << 20))
    fixed = 2 if rng.random() < 0.15 else 0
    chunk = int(rng.choice([8, 16, 40, 100]))
    ctx = dict(trial=trial, arch=arch, arith=arith, mode=mode, B=B, L=L, lens=lens.tolist(), seed=seed, fixed=fixed, chunk=chunk)
    m = model(arch, arith)
    model(arch, "f32", ref=True)
    unsplit = model(arch, arith, ref="unsplit")
    ggml = bool(rng.random() < 0.2)
    tables(arch, ggml)
    ctx["ggml_tables"] = ggml
    kw = dict(id_lengths=lens, mode=mode, noise_seed=seed, fixed_duration=fixed)
    A = m.process_batch(ids, **kw)
    assert all(np.isfinite(p).all() for p in A[0]), ("non-finite pcm", ctx)
    # a row alone
    for b in rng.permutation(B)[:2]:
        b = int(b)
        one = m.process_batch(ids[b:b + 1, :lens[b]], mode=mode, noise_seed=seed + b, fixed_duration=fixed)
        same(([A[0][b]], A[1][b:b + 1], A[2][b:b + 1]), one, "row alone", ctx)
        counts["single"] += 1
    # the call split in two pipelined parts (this handle, B >= 2) == the unsplit call
    if B >= 2 or arith in ("f16", "bf16", "f32"):
        same(A, unsplit.process_batch(ids, **kw), "split in two / segments of tiles", ctx)
        counts["split"] += 1
    # windowed vocoder
    same(A, m.process_batch(ids, vocoder_chunk_frames=chunk, **kw), "windowed", ctx)
    counts["windowed"] += 1
    # pipelined, twice in flight
    m.submit_batch(ids, id_lengths=lens, mode=mode, noise_seed=seed, fixed_duration=fixed)
    m.submit_batch(ids, id_lengths=lens, mode=mode, noise_seed=seed, fixed_duration=fixed, vocoder_chunk_frames=chunk if rng.random() < 0.5 else 0)
    same(A, m.wait(), "pipelined 1", ctx)
    same(A, m.wait(), "pipelined 2", ctx)
    counts["pipelined"] += 1
    if arith != "f32":
        A32 = model(arch, "f32", ref=True).process_batch(ids, **kw)
        assert np.array_equal(A[1], A32[1]) and np.array_equal(A[2], A32[2]), ("16-bit durations differ from fp32", ctx)
        counts["dur16"] += 1
    if oracle is not None and arch == pkg.SYNTH_TINY and L <= 64 and not ggml:  # (the emulated tables are not bit-identical between host and device by construction)
        if arch not in oracles:
            oracles[arch] = oracle.Model(pkg.synth_model_bytes(0x5EED, arch))
        b = int(rng.integers(B))
        ref = oracles[arch].process_ids(ids[b, :lens[b]], mode=mode, noise_kind=oracle.NOISE_COUNTER, noise_seed=seed + b, fixed_duration=fixed,
                                        arith=ORACLE_ARITH[arith], taps=("waveform", "durations"))
        assert ref["waveform"].size == A[1][b], ("sample count vs oracle", ctx, ref["waveform"].size, A[1][b])
        assert int(ref["durations"].sum()) == A[2][b], ("frames vs oracle", ctx)
        e = rel_err(A[0][b], ref["waveform"])
        worst = max(worst, e)
        tol = {"f32": 1e-4, "f32split": 1e-4, "f16": 5e-3, "bf16": 8e-2}[arith]  # (tests/test_gpu_arith16.py: where a 16-bit rounding flips, the error is one ulp of the 16-bit type)
        if e >= tol:
            d = np.abs(A[0][b].astype(np.float64) - ref["waveform"].astype(np.float64))
            rms = np.sqrt((ref["waveform"].astype(np.float64) ** 2).mean())
            print("waveform vs oracle above tolerance:", e, "rms error / rms %.2e" % (np.sqrt((d ** 2).mean()) / rms), "samples above tol/2: %d of %d" % (int((d > tol / 2 * rms).sum()), d.size),
                  "argmax", int(d.argmax()), ctx, flush=True)
            # a 16-bit rounding that flips between the two implementations moves a handful of samples by up to a few 16-bit ulps of an
            # activation; anything systematic shows in the RMS
            assert arith not in ("f32", "f32split") and e < 3 * tol and np.sqrt((d ** 2).mean()) / rms < tol / 5, ("waveform vs oracle", e, ctx)
            counts["oracle_isolated_flips"] = counts.get("oracle_isolated_flips", 0) + 1
        counts["oracle"] += 1
    counts["trials"] += 1
for e in models.values():
    e[0].close()
print("fuzz ok:", counts, "worst waveform error vs oracle %.2e x RMS" % worst, "%.1f s" % (time.time() - t0))
