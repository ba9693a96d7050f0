#!/bin/bash
# streaming whole-resblock kernel (VITS_RBB_STREAM_TILES): bit-identity against the one-tile form, then the A/B at batch 64 x 128 ids
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_stream; mkdir -p $O
for a in f16 bf16; do
  echo "== identity $a" 
  VITS_KNOB_ARITH=$a python3 tools/knob_identity.py
  for t in 2 3 8; do VITS_KNOB_ARITH=$a VITS_RBB_STREAM_MIN_BLOCKS=1 VITS_RBB_STREAM_TILES=$t python3 tools/knob_identity.py; done
  VITS_KNOB_ARITH=$a VITS_RBB_STREAM_MIN_BLOCKS=1 VITS_RBB_STREAM_TILES=5 VITS_RBB_C64K11=1 python3 tools/knob_identity.py
done 2>&1 | tee $O/identity.txt
python3 - <<'PY' 2>&1 | tee $O/identity64.txt
import os, sys, hashlib
import numpy as np
sys.path.insert(0, "tests")
from conftest import load_package
pkg = load_package()
data = pkg.synth_model_bytes(0x5EED, 0)
ids = pkg.synth_ids(24, 128)
lens = np.array([128 - 5 * (i % 7) for i in range(24)], np.int32)
hs = []
for cfg in ({"VITS_RBB_STREAM_TILES": "0"}, {}, {"VITS_RBB_STREAM_TILES": "3", "VITS_RBB_STREAM_MIN_BLOCKS": "16"}, {"VITS_RBB_C64K11": "1"}):
    os.environ.update(cfg)
    m = pkg.Model(data)
    for k in cfg: del os.environ[k]
    m.set_arith(pkg.ARITH_F16)
    pcm, l, f = m.process_batch(ids, id_lengths=lens, noise_seed=3)
    h = hashlib.sha256()
    for p in pcm: h.update(p.tobytes())
    hs.append(h.hexdigest()[:16]); print(cfg, hs[-1], flush=True)
    m.close()
print("IDENTICAL" if len(set(hs)) == 1 else "DIFFERENT")
PY
python3 tools/bN_knobs.py "64" "VITS_RBB_STREAM_TILES=0" "VITS_RBB_STREAM_TILES=2" "VITS_RBB_STREAM_TILES=4" "VITS_RBB_STREAM_TILES=8" "VITS_RBB_STREAM_TILES=16" "VITS_RBB_STREAM_TILES=8 VITS_RBB_STREAM_MIN_BLOCKS=768" "VITS_RBB_STREAM_TILES=8 VITS_RBB_C64K11=1" "VITS_RBB_STREAM_TILES=0 VITS_RBB_C64K11=1" "VITS_RBB_STREAM_TILES=0" 2>&1 | tee $O/ab64.txt
