"""Full-size drill of the model-file path against the REFERENCE'S OWN exporter (build container only, CPU): a default-config
transformers.VitsModel (the MMS-TTS architecture, 36.3 M parameters) goes through /root/reference/scripts/export_vits.py:5-93 into a
temporary file, and that file — 698 tensors, within the 80 bytes of the checkpoint the reference ships as an LFS pointer
(/root/reference/scripts/vits-english.ggml: 74,551,853 B; the difference is the tokenizer's vocabulary strings) — passes the product's
host-only load check (vits_model_file_validate == everything Engine::load checks: format, hyper-parameters, every tensor the engine asks
for with its shape and dtype), survives the product's reader / writer byte for byte, and loads and runs in the oracle. Nothing here
travels to the GPU box (the reference tree does not exist there: the test skips); the tiny exporter-written fixture
tests/golden/tiny_hf_export.ggml covers the GPU side. VERDICT r4 next 8."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

REF_SCRIPTS = "/root/reference/scripts"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF_SCRIPTS, "export_vits.py")), reason="reference tree not present (GPU box)")


@pytest.fixture(scope="module")
def exported(tmp_path_factory):
    torch = pytest.importorskip("torch")
    transformers = pytest.importorskip("transformers")
    sys.path.insert(0, REF_SCRIPTS)
    try:
        import export_vits  # import-safe: __main__ guard at export_vits.py:95
    finally:
        sys.path.pop(0)
    torch.manual_seed(3)
    model = transformers.VitsModel(transformers.VitsConfig()).eval()
    model = export_vits.remove_weight_norm_and_convert_to_fp16(model)

    class Tok:  # what serialize_model_to_binary reads of a tokenizer (export_vits.py:8-27); the MMS-TTS English vocabulary has 38 entries
        phonemize = False
        is_uroman = False
        add_blank = True
        normalize = True
        pad_token = "<pad>"
        unk_token = "<unk>"

        def get_vocab(self):
            v = {"<pad>": 0, " ": 1, "'": 2, "-": 3}
            v.update({c: 4 + i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")})
            v.update({c: 30 + i for i, c in enumerate("0123456")})
            v["<unk>"] = 37
            return v

    path = tmp_path_factory.mktemp("export") / "vits-default.ggml"
    with contextlib.redirect_stdout(io.StringIO()):
        export_vits.serialize_model_to_binary(model.config, model.state_dict(), Tok(), str(path))
    data = path.read_bytes()
    path.unlink()
    return data


def test_default_config_export_matches_the_shipped_checkpoints_size_and_manifest(exported):
    from modelfile_py import parse_model_file
    with open(os.path.join(REF_SCRIPTS, "vits-english.ggml")) as fh:  # an LFS pointer: three text lines
        shipped = int([ln.split()[1] for ln in fh if ln.startswith("size ")][0])
    assert shipped == 74551853
    assert abs(len(exported) - shipped) <= 80, (len(exported), shipped)
    f = parse_model_file(exported)
    t = f["tensors"]
    assert len(t) == 698
    groups = {}
    for name in t:
        groups[name.split(".")[0]] = groups.get(name.split(".")[0], 0) + 1
    assert groups == {"duration_predictor": 284, "decoder": 155, "text_encoder": 111, "flow": 80, "posterior_encoder": 68}
    # spot checks of App. C's examples: ggml ne = reversed torch shape; conv weights fp16 (type 1), the rest fp32
    w, dt = t["flow.flows.0.wavenet.in_layers.0.weight"]
    assert w.shape == (384, 192, 5) and dt == 1
    w, dt = t["decoder.upsampler.0.weight"]
    assert w.shape == (512, 256, 16) and dt == 1
    w, dt = t["text_encoder.encoder.layers.0.attention.emb_rel_k"]
    assert w.shape == (1, 9, 96) and dt == 0
    w, dt = t["duration_predictor.flows.0.log_scale"]
    assert w.shape == (2, 1) and dt == 0
    assert f["config"]["upsample_rates"] == "[8, 8, 2, 2]" and f["config"]["use_stochastic_duration_prediction"] == "True"
    assert len(f["vocab"]) == 38 and f["add_blank"] == 1


def test_product_reader_accepts_and_round_trips_the_full_size_export(pkg, exported):
    # host-only: no device is touched (Engine::validate runs load()'s checks in dry-run mode)
    pkg.validate(exported)
    assert pkg.reserialize(exported) == exported
    # and a file with one engine-side tensor renamed is rejected by name
    bad = exported.replace(b"decoder.resblocks.11.convs2.2.weight", b"decoder.resblocks.11.convs2.2.weighx")
    assert bad != exported
    with pytest.raises(pkg.VitsError, match="decoder.resblocks.11.convs2.2.weight"):
        pkg.validate(bad)
    ids = pkg.file_tokenize(exported, "Hello world")
    assert ids.size == 2 * 11 + 1 and (ids[::2] == 0).all()  # blanks interspersed (src/vits_tokenizer.cpp:200-208)


def test_oracle_loads_and_runs_the_full_size_export(exported):
    import oracle_lib as O
    om = O.Model(exported)
    ids = np.array([0, 9, 0, 17, 0], np.int32)
    r = om.process_ids(ids, mode=O.MODE_REFERENCE, noise_kind=O.NOISE_COUNTER, noise_seed=1, threads=4, taps=["waveform", "durations"])
    assert r["durations"].size == 5 and (r["durations"] >= 1).all()
    assert r["waveform"].size == 256 * int(r["durations"].sum()) + 294 and np.isfinite(r["waveform"]).all()  # Q1: no crop in reference mode
