"""Test helper (it runs the oracle, so it lives under tests/): where does a 16-bit-mode run differ from the oracle in the same arithmetic? (long utterance, per-tap statistics,
the largest deviations, and the same comparison with the fused pair kernel / the group-layout vocoder switched off)."""
import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from conftest import load_package
    import oracle_lib as O
    pkg = load_package()
    fb = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL)
    T = int(os.environ.get("DIAG_T", "1024"))
    ids = pkg.synth_ids(1, T, ids_seed=77)[0]
    om = O.Model(fb)
    m = pkg.Model(fb)
    m.set_arith(pkg.ARITH_F16)
    g, _, _ = m.process_batch(ids, noise_seed=12, fixed_duration=2, collect_taps=True)
    ref = om.process_ids(ids, noise_kind=O.NOISE_COUNTER, noise_seed=12, fixed_duration=2, arith=O.ARITH_F16)
    a, b = m.tap("pre_tanh").astype(np.float64), ref["pre_tanh"].astype(np.float64)
    d = np.abs(a - b); rms = np.sqrt((b ** 2).mean())
    top = np.argsort(d)[-12:][::-1]
    print(os.environ.get("TAG", ""), "pre_tanh max %.3e p99.9 %.3e rms %.3e" % (d.max() / rms, np.percentile(d, 99.9) / rms, np.sqrt((d ** 2).mean()) / rms))
    print("   worst samples:", [(int(i), round(float(d[i] / rms), 4)) for i in top])
    # error vs position modulo the block widths of the last stage (fused pair: 246 / 250 / 254 output columns; conv16: 256)
    big = np.nonzero(d > 10 * np.sqrt((d ** 2).mean()))[0]
    print("   samples with error > 10 rms-error:", big.size, "first", big[:10].tolist())
    sys.exit(0)
for tag, env in (("default", {}), ("no fused pairs", {"VITS_NO_FUSE16": "1"}), ("no group layout", {"VITS_NO_GROUP16": "1"})):
    e = dict(os.environ, TAG=tag, **env)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e)
