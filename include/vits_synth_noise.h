/*
 * vits_synth_noise.h — definition of the SYNTHETIC input streams (counter-based noise and the
 * synthetic id generator) used by the benchmark and the parity tests.
 *
 * This is an input DATA definition (like a file format), shared by the product (host + device
 * code) and by the oracle, so that both sides see bit-identical inputs. It contains no model
 * arithmetic. Everything is integer hashing plus ONE float multiply, so host (g++) and device
 * (hipcc) produce bit-identical floats.
 *
 * The reference draws its noise from a process-global libstdc++ engine
 * (/root/reference/src/vits.cpp:31, src/include/ggml-util.h:187-199: std::default_random_engine +
 * std::normal_distribution<float>, drawn sequentially on the host). That stream is kept as
 * noise kind VITS_NOISE_REFERENCE for the batch-1 drop-in call; it is inherently serial, so the
 * batched/benchmark path uses the counter-based stream below (noise kind VITS_NOISE_COUNTER),
 * which any thread can evaluate at any index.
 */
#ifndef VITS_SYNTH_NOISE_H
#define VITS_SYNTH_NOISE_H

#include <stdint.h>

#if defined(__HIPCC__) /* compiled by hipcc: usable from gfx950 device code; plain host C/C++ otherwise (oracle, tests) */
#define VITS_HD __host__ __device__ static inline
#else
#define VITS_HD static inline
#endif

/* splitmix64 finalizer */
VITS_HD uint64_t vits_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* 64-bit hash of (seed, stream, index) */
VITS_HD uint64_t vits_hash3(uint64_t seed, uint64_t stream, uint64_t index) {
    uint64_t h = vits_mix64(seed ^ 0xD1B54A32D192ED03ull);
    h = vits_mix64(h ^ (stream * 0x9E3779B97F4A7C15ull));
    h = vits_mix64(h ^ index);
    return h;
}

/*
 * Approximately N(0,1): Irwin-Hall sum of eight 16-bit uniforms (two 64-bit hashes), centred and
 * scaled. Exact integer sum -> one exact int->float conversion, one exact subtraction, ONE rounded
 * multiply: bit-identical on every IEEE-754 machine. Range +-4.9 sigma.
 * mean = 8 * 65535/2 = 262140 ; var = 8 * (65536^2 - 1)/12 -> std = 53509.536...
 */
VITS_HD float vits_counter_normal(uint64_t seed, uint64_t stream, uint64_t index) {
    uint64_t a = vits_hash3(seed, stream, 2 * index);
    uint64_t b = vits_hash3(seed, stream, 2 * index + 1);
    uint32_t s = (uint32_t)(a & 0xFFFF) + (uint32_t)((a >> 16) & 0xFFFF) + (uint32_t)((a >> 32) & 0xFFFF) +
                 (uint32_t)((a >> 48) & 0xFFFF) + (uint32_t)(b & 0xFFFF) + (uint32_t)((b >> 16) & 0xFFFF) +
                 (uint32_t)((b >> 32) & 0xFFFF) + (uint32_t)((b >> 48) & 0xFFFF);
    float centred = (float)(int32_t)s - 262140.0f;
    return centred * 1.8688258e-05f; /* 1 / 53509.536 */
}

/* uniform integer in [0, n) */
VITS_HD uint32_t vits_counter_uniform(uint64_t seed, uint64_t stream, uint64_t index, uint32_t n) {
    return (uint32_t)((vits_hash3(seed, stream, index) >> 33) % n);
}

/* stream ids */
#define VITS_STREAM_NOISE_DUR 1u   /* duration-predictor latents, index = c*T + t (c in {0,1})      */
#define VITS_STREAM_NOISE_PRIOR 2u /* prior noise, index = c*L + t (c in [0,192))                   */
#define VITS_STREAM_IDS 3u         /* synthetic ids, index = t                                      */
#define VITS_STREAM_WEIGHTS 16u    /* synthetic weights: stream = 16 + tensor ordinal, index = elem */

/*
 * Synthetic ids for utterance `utt`: even positions 0 (blank), odd positions
 * uniform in {1..vocab-1}; seed = ids_seed + utt.
 */
VITS_HD int32_t vits_synth_id(uint64_t ids_seed, uint32_t utt, uint32_t t, uint32_t vocab) {
    if ((t & 1u) == 0u) return 0;
    return 1 + (int32_t)vits_counter_uniform(ids_seed + utt, VITS_STREAM_IDS, t, vocab - 1);
}

#endif /* VITS_SYNTH_NOISE_H */
