// model_file.h — the reference's on-disk model format, host side.
//
// Reader restates /root/reference/src/vits_tokenizer.cpp:22-55 (tokenizer block) and
// /root/reference/src/vits_model_data.cpp:29-97 (config + tensor blocks); writer restates
// /root/reference/scripts/export_vits.py:5-70. All integers are little-endian u32.
//
//   u32 vocab_size; repeat{ u32 len; bytes key; u32 id }
//   u32 add_blank; u32 normalize; u32 len; bytes pad_token; u32 len; bytes unk_token
//   u32 n_cfg;     repeat{ u32 len; bytes key; u32 len; bytes value }
//   u32 n_tensors; repeat{ u32 len; bytes name; u32 type(0=f32,1=f16[,2=bf16 extension]); u32 rank;
//                          u32 dim[rank] (reversed torch shape == ggml ne); u32 nbytes; payload }
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace vits {

enum : uint32_t { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 /* extension, SURVEY.md §8f rank 1 */ };

struct TensorEntry {
    std::string name;
    uint32_t dtype = DT_F32;
    uint32_t rank = 0;
    int64_t ne[4] = {1, 1, 1, 1};  // file order (fastest dim first)
    std::vector<uint8_t> raw;      // payload exactly as stored
    int64_t count() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
    std::vector<float> to_f32() const;  // widen
};

struct ModelFile {
    std::vector<std::pair<std::string, uint32_t>> vocab;  // file order preserved (byte-exact round trip)
    uint32_t add_blank = 1, normalize = 1;
    std::string pad_token, unk_token;
    std::vector<std::pair<std::string, std::string>> config;  // file order preserved
    std::vector<TensorEntry> tensors;

    // parse; on failure returns false and sets err
    bool parse(const uint8_t* bytes, size_t size, std::string& err);
    std::vector<uint8_t> serialize() const;

    const TensorEntry* find(const std::string& name) const;
    std::string cfg(const std::string& key, const std::string& dflt = "") const;

  private:
    mutable std::map<std::string, size_t> index_;
};

uint16_t f32_to_f16(float f);  // round-to-nearest-even, like torch .to(float16) (export_vits.py:87)
float f16_to_f32(uint16_t h);
uint16_t f32_to_bf16(float f);
float bf16_to_f32(uint16_t h);

// Hyper-parameters, parsed ONCE at load (the reference re-parses the strings on every graph build,
// vits.cpp:33-110; keys read at :246-254,453-457,501,523,585-595,648-649,858-861,930,977-979).
// Missing keys fall back to transformers.VitsConfig defaults (== facebook/mms-tts-*).
struct HParams {
    int vocab_size = 38, hidden = 192, layers = 6, heads = 2, window = 4, ffn_dim = 768, ffn_k = 3, flow_size = 192;
    int n_flows = 4, wn_layers = 4, wn_k = 5, wn_rate = 1;
    int up_init = 512;
    std::vector<int> up_rates{8, 8, 2, 2}, up_k{16, 16, 4, 4}, rb_k{3, 7, 11};
    std::vector<std::vector<int>> rb_d{{1, 3, 5}, {1, 3, 5}, {1, 3, 5}};
    float lrelu = 0.1f, ln_eps = 1e-5f;
    int dp_k = 3, dds_layers = 3, dp_bins = 10, dp_flows = 4;
    float dp_tail = 5.f, noise_scale_dur = 0.8f, noise_scale = 0.667f, speaking_rate = 1.0f;
    int sampling_rate = 16000;
    std::string hidden_act = "relu";
    bool stochastic_duration = true;
    int speaker_embedding_size = 0;
    bool load(const ModelFile& f, std::string& err);
};

// Synthetic model files (include/vits.h: vits_synth_model_bytes).
ModelFile make_synthetic_model(uint64_t seed, int arch);

}  // namespace vits
