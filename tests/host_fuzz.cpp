// host_fuzz.cpp — TEST INFRASTRUCTURE: the product's host code that eats untrusted bytes (vits_model_load_from_bytes' parser, hyper-parameter reader,
// per-tensor shape validation, packing, tokenizer, writer: model_file.cpp, engine_load.cpp, engine_support.cpp, abi.cpp) under AddressSanitizer + UBSan and a
// seeded mutation fuzzer (only the oracle ever saw a sanitizer). Linked against the SANITIZED build of the product's host translation
// units (vits.cpp_amd/csrc/Makefile target `asan`), driven through the C ABI only — the entry points a caller hands a file to:
//   vits_model_file_validate (= everything Engine::load checks and packs, without a device), vits_model_file_reserialize, vits_model_file_tokenize.
// Mutations of a real exporter-written file (tests/golden/tiny_hf_export.ggml): truncation at every kind of place, byte flips, and extreme values in
// every u32 of the container (vocabulary size, string lengths, config count, tensor count, type, rank, dims, byte length) — the reader the reference trusts
// blindly (/root/reference/src/vits_model_data.cpp:29-97, /root/reference/src/vits_tokenizer.cpp:22-78). Any sanitizer report aborts the process.
// usage: host_fuzz FILE [mutations = 20000] [seed = 1]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/vits.h"

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

// offsets of every u32 field of the container, found by walking a WELL-FORMED file
static std::vector<size_t> header_fields(const std::vector<uint8_t>& f) {
    std::vector<size_t> at;
    size_t off = 0;
    auto u32 = [&]() -> uint32_t {
        uint32_t v = 0;
        if (off + 4 <= f.size()) std::memcpy(&v, f.data() + off, 4);
        at.push_back(off);
        off += 4;
        return v;
    };
    auto str = [&] { off += u32(); };
    const uint32_t nv = u32();
    for (uint32_t i = 0; i < nv && off < f.size(); ++i) {
        str();
        u32();
    }
    u32(), u32();
    str(), str();
    const uint32_t nc = u32();
    for (uint32_t i = 0; i < nc && off < f.size(); ++i) str(), str();
    const uint32_t nt = u32();
    for (uint32_t i = 0; i < nt && off < f.size(); ++i) {
        str();
        u32();
        const uint32_t rank = u32();
        for (uint32_t r = 0; r < rank && r < 8; ++r) u32();
        off += u32();
    }
    if (off != f.size()) {
        std::fprintf(stderr, "host_fuzz: the seed file is not well-formed (walk ended at %zu of %zu)\n", off, f.size());
        std::exit(2);
    }
    return at;
}

struct Tally {
    long validated = 0, rejected = 0, reserialized = 0, tokenized = 0;
};

static void drive(const std::vector<uint8_t>& m, Tally& t) {
    const char* p = reinterpret_cast<const char*>(m.data());
    if (vits_model_file_validate(p, m.size()) == 0) ++t.validated;
    else {
        ++t.rejected;
        if (!vits_last_error() || !*vits_last_error()) {
            std::fprintf(stderr, "host_fuzz: a rejection without a message\n");
            std::exit(3);
        }
    }
    char* out = nullptr;
    size_t n = 0;
    if (vits_model_file_reserialize(p, m.size(), &out, &n) == 0) {
        ++t.reserialized;
        // what parses is written back and parses to the same bytes again (the writer is the reader's inverse)
        char* out2 = nullptr;
        size_t n2 = 0;
        if (vits_model_file_reserialize(out, n, &out2, &n2) != 0 || n2 != n || std::memcmp(out, out2, n) != 0) {
            std::fprintf(stderr, "host_fuzz: reserialize is not idempotent on a file it accepted\n");
            std::exit(4);
        }
        vits_free_bytes(out2);
        vits_free_bytes(out);
    }
    int32_t ids[64];
    if (vits_model_file_tokenize(p, m.size(), "Hello, World! 0123 \xc3\xa9\xff", ids, 64) >= 0) ++t.tokenized;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const long n_mut = argc > 2 ? std::atol(argv[2]) : 20000;
    rng_state ^= (uint64_t)(argc > 3 ? std::atol(argv[3]) : 1) * 0x9E3779B97F4A7C15ull;
    std::vector<uint8_t> base;
    {
        FILE* f = std::fopen(argv[1], "rb");
        if (!f) return 2;
        uint8_t buf[65536];
        size_t k;
        while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) base.insert(base.end(), buf, buf + k);
        std::fclose(f);
    }
    const std::vector<size_t> fields = header_fields(base);
    Tally t;
    drive(base, t);
    if (t.validated != 1 || t.reserialized != 1 || t.tokenized != 1) {
        std::fprintf(stderr, "host_fuzz: the unmutated file was not accepted: %s\n", vits_last_error());
        return 5;
    }
    // the synthetic files of the product's own writer round-trip byte for byte and validate
    for (int arch : {VITS_SYNTH_TINY, VITS_SYNTH_TINY | VITS_SYNTH_BF16}) {
        char* sb = nullptr;
        size_t sn = 0;
        if (vits_synth_model_bytes(0x5EED + arch, arch, &sb, &sn) != 0) return 6;
        char* rb = nullptr;
        size_t rn = 0;
        if (vits_model_file_reserialize(sb, sn, &rb, &rn) != 0 || rn != sn || std::memcmp(sb, rb, sn) != 0 || vits_model_file_validate(sb, sn) != 0) {
            std::fprintf(stderr, "host_fuzz: synthetic file (arch %d) does not round-trip: %s\n", arch, vits_last_error());
            return 6;
        }
        vits_free_bytes(rb);
        vits_free_bytes(sb);
    }
    static const uint32_t extreme[] = {0u, 1u, 2u, 3u, 7u, 8u, 255u, 256u, 65535u, 65536u, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu, 0x10000000u, 0x00ffffffu};
    for (long i = 0; i < n_mut; ++i) {
        std::vector<uint8_t> m = base;
        const int kind = (int)(i & 3);
        if (kind == 0) {  // truncation: anywhere, and right behind / inside a header field
            size_t cut = rnd() % (m.size() + 1);
            if (i & 4) cut = std::min(m.size(), fields[rnd() % fields.size()] + (size_t)(rnd() % 6));
            m.resize(cut);
        }
        if (kind == 1 || kind == 3) {  // byte flips, biased towards the header region of the tensor block and the front of the file
            const int nflip = 1 + (int)(rnd() % 8);
            const bool in_header = (rnd() & 3) != 0;  // one mutation in four leaves the container alone (payload / strings only: mostly still a loadable file)
            for (int k = 0; k < nflip; ++k) {
                size_t at = rnd() % m.size();
                if (in_header && (rnd() & 1)) at = std::min(m.size() - 1, fields[rnd() % fields.size()] + (size_t)(rnd() % 4));
                m[at] ^= (uint8_t)(1u << (rnd() % 8));
            }
        }
        if (kind == 2 || kind == 3) {  // an extreme (or off-by-one) value in one of the container's u32 fields
            const size_t at = fields[rnd() % fields.size()];
            uint32_t v;
            std::memcpy(&v, m.data() + at, 4);
            const uint64_t r = rnd();
            if (r % 3 == 0) v = v + (uint32_t)((r >> 8) % 5) - 2u;
            else if (r % 3 == 1) v = v * 2u + (uint32_t)((r >> 8) & 1);
            else v = extreme[(r >> 8) % (sizeof extreme / sizeof extreme[0])];
            std::memcpy(m.data() + at, &v, 4);
            if ((r >> 20) % 7 == 0 && m.size() > 16) m.resize(m.size() - 1 - (size_t)((r >> 24) % 16));
        }
        drive(m, t);
    }
    std::printf("host_fuzz ok: %ld mutations of %zu bytes (%zu header fields): %ld validated, %ld rejected, %ld reserialized, %ld tokenized\n", n_mut, base.size(), fields.size(),
                t.validated - 1, t.rejected, t.reserialized - 1, t.tokenized - 1);
    return 0;
}
