"""GPU parity of each HIP operator against the oracle's restatement of the same reference lines (through the C ABI).
fp32 tolerance: max |gpu - oracle| <= 2e-5 * RMS(oracle) per operator (summation order differs: MFMA k-order vs the
oracle's loop order); integer/index results exact."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("cin,cout,k,dil,T", [
    (32, 32, 3, 1, 300), (32, 32, 7, 3, 517), (32, 32, 11, 5, 1000), (64, 64, 3, 3, 256), (64, 64, 11, 1, 700),
    (128, 128, 7, 5, 400), (256, 256, 3, 1, 130), (256, 256, 11, 5, 260), (192, 768, 3, 1, 128), (768, 192, 3, 1, 128),
    (192, 512, 7, 1, 77), (192, 192, 1, 1, 128), (192, 29, 1, 1, 50), (96, 192, 1, 1, 33), (16, 32, 3, 1, 20), (5, 7, 5, 2, 19),
    # 16-byte aligned rows (dwordx4 LDS-DMA path, tile origin shifted by 0..3) with lengths just past tile / float4 boundaries
    (128, 128, 11, 5, 132), (64, 64, 7, 1, 1028), (256, 128, 3, 5, 516), (32, 64, 11, 3, 260), (128, 256, 3, 1, 4), (64, 32, 7, 3, 8),
])
def test_conv1d_matches_oracle(pkg, oracle, cin, cout, k, dil, T):
    rng = np.random.default_rng(cin * 1000 + cout + k)
    x = rnd(rng, 2, cin, T)
    w = rnd(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
    b = rnd(rng, cout, scale=0.1)
    lens = np.array([T, max(1, T - 37)], np.int32)
    y_gpu = pkg.op_conv1d(x, w, b, dilation=dil, lens=lens)
    y_ref = oracle.conv1d(x, w, b, dilation=dil, lens=lens)
    for i in range(2):
        assert rel_err(y_gpu[i, :, :lens[i]], y_ref[i, :, :lens[i]]) < TOL


def test_conv1d_fused_epilogues(pkg, oracle):
    rng = np.random.default_rng(5)
    cin, cout, k, T = 64, 64, 7, 333
    x, w, b = rnd(rng, 3, cin, T), rnd(rng, cout, cin, k, scale=0.05), rnd(rng, cout, scale=0.1)
    res, acc = rnd(rng, 3, cout, T), rnd(rng, 3, cout, T)
    lens = np.array([333, 100, 1], np.int32)
    for kw in [dict(pre_slope=0.1), dict(pre_slope=0.1, residual=res), dict(pre_slope=0.1, residual=res, accum=acc, out_scale=1.0 / 3.0),
               dict(post_act=1), dict(pad_left=1)]:
        if "pad_left" in kw:
            w3 = w[:, :, :3].copy()
            yg = pkg.op_conv1d(x, w3, b, lens=lens, **kw)
            yr = oracle.conv1d(x, w3, b, lens=lens, **kw)
        else:
            yg = pkg.op_conv1d(x, w, b, lens=lens, **kw)
            yr = oracle.conv1d(x, w, b, lens=lens, **kw)
        for i in range(3):
            assert rel_err(yg[i, :, :lens[i]], yr[i, :, :lens[i]]) < TOL, kw


@pytest.mark.parametrize("half,T", [(192, 200), (16, 70)])
def test_gated_conv_matches_oracle(pkg, oracle, half, T):
    rng = np.random.default_rng(9)
    x = rnd(rng, 2, half, T)
    w = rnd(rng, 2 * half, half, 5, scale=1.0 / np.sqrt(half * 5))
    b = rnd(rng, 2 * half, scale=0.1)
    lens = np.array([T, T - 11], np.int32)
    yg = pkg.op_conv1d(x, w, b, post_act=2, lens=lens)
    yr = oracle.conv1d(x, w, b, post_act=2, lens=lens)
    assert yg.shape == (2, half, T)
    for i in range(2):
        assert rel_err(yg[i, :, :lens[i]], yr[i, :, :lens[i]]) < TOL


@pytest.mark.parametrize("cin,cout,k,s,T", [(512, 256, 16, 8, 40), (256, 128, 16, 8, 90), (128, 64, 4, 2, 300), (64, 32, 4, 2, 515), (32, 16, 8, 4, 21)])
@pytest.mark.parametrize("crop_mode", ["hf", "reference"])
def test_conv_transpose_matches_oracle(pkg, oracle, cin, cout, k, s, T, crop_mode):
    rng = np.random.default_rng(cin + k)
    x = rnd(rng, 2, cin, T)
    w = rnd(rng, cin, cout, k, scale=1.0 / np.sqrt(cin * 2))
    b = rnd(rng, cout, scale=0.1)
    crop = (k - s) // 2 if crop_mode == "hf" else 0  # Q1: the reference never crops (vits.cpp:187)
    lens = np.array([T, T - 5], np.int32)
    yg = pkg.op_conv_transpose1d(x, w, b, s, crop, pre_slope=0.1, lens=lens)
    yr = oracle.conv_transpose1d(x, w, b, s, crop, pre_slope=0.1, lens=lens)
    for i in range(2):
        lo = s * lens[i] + k - s - 2 * crop
        assert rel_err(yg[i, :, :lo], yr[i, :, :lo]) < TOL


@pytest.mark.parametrize("T", [3, 5, 16, 17, 128, 257, 300, 1100, 2049])
def test_rel_attention_matches_oracle(pkg, oracle, T):
    rng = np.random.default_rng(T)
    heads, hd, w = 2, 96, 4
    q, k, v = rnd(rng, 2, heads * hd, T, scale=0.3), rnd(rng, 2, heads * hd, T, scale=0.3), rnd(rng, 2, heads * hd, T)
    rk, rv = rnd(rng, 2 * w + 1, hd, scale=0.1), rnd(rng, 2 * w + 1, hd, scale=0.1)
    lens = np.array([T, max(1, T - 2)], np.int32)
    og = pkg.op_rel_attention(q, k, v, rk, rv, heads, w, lens=lens)
    orf = oracle.rel_attention(q, k, v, rk, rv, heads, w, lens=lens)
    for i in range(2):
        assert rel_err(og[i, :, :lens[i]], orf[i, :, :lens[i]]) < TOL


@pytest.mark.parametrize("w", [0, 1, 7, 8, 10, 33])
def test_rel_attention_other_window_sizes_match_oracle(pkg, oracle, w):
    """window_size is a hyper-parameter of the model file (vits.cpp:246-254): 2w+1 relative positions, more than one 16-column tile of
    the q.Ek product from w = 8 on (the matrix-core kernel used to fill only the first 16 columns of its table)."""
    rng = np.random.default_rng(100 + w)
    heads, hd = 2, 48
    for T in (5, 70, 300):
        q, k, v = rnd(rng, 2, heads * hd, T, scale=0.3), rnd(rng, 2, heads * hd, T, scale=0.3), rnd(rng, 2, heads * hd, T)
        rk, rv = rnd(rng, 2 * w + 1, hd, scale=0.3), rnd(rng, 2 * w + 1, hd, scale=0.3)
        lens = np.array([T, max(1, T - 3)], np.int32)
        og = pkg.op_rel_attention(q, k, v, rk, rv, heads, w, lens=lens)
        orf = oracle.rel_attention(q, k, v, rk, rv, heads, w, lens=lens)
        for i in range(2):
            assert rel_err(og[i, :, :lens[i]], orf[i, :, :lens[i]]) < TOL, (w, T, i)


@pytest.mark.parametrize("hd", [16, 64, 128])
def test_rel_attention_other_head_sizes_match_oracle(pkg, oracle, hd):
    """head_dim / 16 = the number of 16-channel d tiles of the P.V product (1, 4, 8): the waves of a block take d tiles w and w + NW, and a
    16-byte aligned stride (T = 260) takes the vector loads of V, any other one the scalar loads (same values, same order)."""
    rng = np.random.default_rng(200 + hd)
    heads, w = 2, 4
    for T in (9, 260, 1300):
        q, k, v = rnd(rng, 2, heads * hd, T, scale=0.3), rnd(rng, 2, heads * hd, T, scale=0.3), rnd(rng, 2, heads * hd, T)
        rk, rv = rnd(rng, 2 * w + 1, hd, scale=0.1), rnd(rng, 2 * w + 1, hd, scale=0.1)
        lens = np.array([T, max(1, T - 5)], np.int32)
        og = pkg.op_rel_attention(q, k, v, rk, rv, heads, w, lens=lens)
        orf = oracle.rel_attention(q, k, v, rk, rv, heads, w, lens=lens)
        for i in range(2):
            assert rel_err(og[i, :, :lens[i]], orf[i, :, :lens[i]]) < TOL, (hd, T, i)


def test_rel_attention_does_not_depend_on_the_longest_member_of_the_batch(pkg):
    """The attention kernel sizes its LDS by the longest utterance of the batch and, for very long ones, takes the V operands of the
    P.V product straight from memory instead of staging them through LDS: a 300-token utterance must come out bit-identical whether it
    runs alone or beside a 2049-token one (same MFMA sequence in both paths)."""
    rng = np.random.default_rng(11)
    heads, hd, w = 2, 96, 4
    # (100 tokens alone: the short-sequence variant — one key tile of look-ahead, four blocks per CU —; 300 alone: four waves, two tiles of
    # look-ahead; beside 2049 tokens: eight waves. 100 and 300 are 16-byte aligned strides, 2049 is not: vector and scalar loads of V.)
    for T, TL in ((300, 2049), (100, 2049), (100, 300)):
        q, k, v = rnd(rng, 1, heads * hd, T, scale=0.3), rnd(rng, 1, heads * hd, T, scale=0.3), rnd(rng, 1, heads * hd, T)
        rk, rv = rnd(rng, 2 * w + 1, hd, scale=0.1), rnd(rng, 2 * w + 1, hd, scale=0.1)
        alone = pkg.op_rel_attention(q, k, v, rk, rv, heads, w, lens=np.array([T], np.int32))

        def pad(a):
            out = rnd(rng, 2, heads * hd, TL, scale=0.3)
            out[0, :, :T] = a[0]
            return out

        both = pkg.op_rel_attention(pad(q), pad(k), pad(v), rk, rv, heads, w, lens=np.array([T, TL], np.int32))
        assert np.array_equal(alone[0, :, :T], both[0, :, :T]), (T, TL)


def test_add_layer_norm_matches_oracle(pkg, oracle):
    rng = np.random.default_rng(3)
    x, r = rnd(rng, 2, 192, 150), rnd(rng, 2, 192, 150)
    g, b = rnd(rng, 192) + 1, rnd(rng, 192)
    assert rel_err(pkg.op_add_layer_norm(x, r, g, b), oracle.add_layer_norm(x, r, g, b)) < TOL


def test_mfma_16x16x4_is_the_same_sequential_fmaf_chain_as_32x32x2(tmp_path):
    """The hardware fact conv_lat16_kernel stands on (small grids): a K chain of v_mfma_f32_16x16x4_f32 is bit for bit the chain of
    v_mfma_f32_32x32x2_f32 and the scalar fmaf chain in k order — tools/mfma_bits.hip, built and run here (1,024 outputs x 256 products);
    the 16-bit pair (32x32x16 / 16x16x32) likewise (tools/mfma_bits16.hip)."""
    import os, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for src, want in (("mfma_bits.hip", "32x32x2 vs 16x16x4: 0 of 1024 differ; 32x32x2 vs fmaf chain: 0; 16x16x4 vs fmaf chain: 0"),
                      ("mfma_bits16.hip", "32x32x16 vs 16x16x32: 0 of 1024 outputs differ")):
        exe = str(tmp_path / src.replace(".hip", ""))
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(root, "tools", src), "-o", exe], check=True, capture_output=True, timeout=300)
        out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
        assert want in out, out
