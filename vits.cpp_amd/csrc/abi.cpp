// abi.cpp — the C ABI of include/vits.h. Nothing below throws across the boundary (the reference lets
// std::runtime_error escape and calls exit(1) from ASSERT: /root/reference/src/vits_model_data.cpp:102,144,
// src/include/debug.h:29-36); failures return NULL / {NULL,0} / -1 and set vits_last_error().
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <vector>

#include "../../include/vits.h"
#include "busy_guard.h"
#include "engine.h"
#include "pcm_gather.h"

namespace vits {
void reference_noise_seed(uint32_t seed);
}

static thread_local std::string g_last_error;
static void set_err(const std::string& e) { g_last_error = e; }

#define VITS_TRY try {
#define VITS_CATCH(ret)                           \
    }                                             \
    catch (const std::exception& e) {             \
        set_err(e.what());                        \
        return ret;                               \
    }                                             \
    catch (...) {                                 \
        set_err("unknown exception");             \
        return ret;                               \
    }

VITS_API const char* vits_last_error(void) { return g_last_error.c_str(); }

// One call at a time per model handle, enforced (busy_guard.h). Distinct handles run concurrently.
#define VITS_ENTER(model, ret)                                                                                                          \
    vits::BusyGuard busy_guard_((model) ? &const_cast<vits_model*>(model)->eng.busy : nullptr);                                        \
    vits::KernelKnobsScope kernel_knobs_scope_((model) ? &(model)->eng.knobs.kernel : nullptr);                                        \
    if ((model) && !busy_guard_.entered()) {                                                                                                 \
        set_err("model busy: another call is in progress on this handle (one call at a time per model; use one handle per thread)"); \
        return ret;                                                                                                                     \
    }

// reference: src/vits.cpp:1205-1215
VITS_API vits_model* vits_model_load_from_bytes(const char* bytes, size_t size) {
    VITS_TRY
    if (!bytes) {
        set_err("null model bytes");
        return nullptr;
    }
    vits_model* m = new vits_model();
    std::string err;
    if (!m->eng.load(reinterpret_cast<const uint8_t*>(bytes), size, err)) {
        set_err(err);
        delete m;
        return nullptr;
    }
    return m;
    VITS_CATCH(nullptr)
}

// reference: src/vits.cpp:1193-1203 -> vits_model_data::from_file src/vits_model_data.cpp:99-109
VITS_API vits_model* vits_model_load_from_file(const char* path) {
    VITS_TRY
    if (!path) {
        set_err("null path");
        return nullptr;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) {
        set_err(std::string("[ERROR] failed to open file: ") + path);  // message of vits_model_data.cpp:102
        return nullptr;
    }
    std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return vits_model_load_from_bytes(buf.data(), buf.size());
    VITS_CATCH(nullptr)
}

// reference: src/vits.cpp:1217-1219
// A handle another thread is inside (or that a callback is running on) is NOT freed: the call would go on using the engine it
// runs on. The refusal path does not touch the flag (the compare-exchange fails without writing: the flag stays the in-flight call's and is
// released when that call returns, so the retry the message asks for succeeds); the success path takes the flag and deletes the handle with it.
VITS_API void vits_free_model(vits_model* model) {
    if (!model) return;
    bool expected = false;
    if (!model->eng.busy.compare_exchange_strong(expected, true, std::memory_order_acquire)) {
        set_err("vits_free_model: model busy (a call is in progress on this handle); not freed — call again when it has returned");
        return;
    }
    try {
        delete model;
    } catch (...) {
    }
}

// reference: src/vits.cpp:1221-1223 (scalar delete on new[] there, Q12; both sides are ours here)
VITS_API void vits_free_result(vits_result result) { delete[] result.data; }

static vits_result process_ids_impl(vits_model* model, const int32_t* ids, size_t n) {
    vits_result r{nullptr, 0};
    if (!model || !ids || n == 0) {
        set_err(n == 0 ? "empty input (no known symbols in the text)" : "null argument");
        return r;
    }
    VITS_ENTER(model, r)
    vits_process_opts o;
    std::memset(&o, 0, sizeof(o));
    o.struct_size = sizeof(o);
    o.mode = VITS_MODE_DEFAULT;
    o.noise_kind = VITS_NOISE_REFERENCE;
    vits_batch_result br;
    std::memset(&br, 0, sizeof(br));
    std::string err;
    const int32_t len = (int32_t)n;
    if (model->eng.process_batch(ids, &len, 1, (int)n, o, &br, err) != 0) {
        set_err(err);
        vits_free_batch_result(&br);
        return r;
    }
    // vits.cpp:1226-1231: a fresh buffer of exactly `size` samples owned by the library
    r.size = (size_t)br.lengths[0];
    if (br.batch == 1 && br.stride >= r.size) {  // one utterance: its row IS the result (no second buffer, no copy)
        r.data = br.data;
        br.data = nullptr;
    } else {
        r.data = new float[r.size];
        std::memcpy(r.data, br.data, sizeof(float) * r.size);
    }
    vits_free_batch_result(&br);
    return r;
}

// reference: src/vits.cpp:1225-1232 -> vits_model::process :1101-1191
VITS_API vits_result vits_model_process(vits_model* model, const char* phonemes) {
    vits_result none{nullptr, 0};
    VITS_TRY
    if (!model || !phonemes) {
        set_err("null argument");
        return none;
    }
    std::vector<int32_t> ids;
    std::string terr;
    if (!model->eng.tok.tokenize_checked(phonemes, ids, terr)) {  // vits.cpp:1109 (a phonetic model: refused, see engine.h Tokenizer)
        set_err(terr);
        return none;
    }
    if (ids.empty() && !model->eng.tok.add_blank) {  // Q11: the reference's tokenizer returns no ids at all without add_blank
        set_err("empty input: the model file says add_blank = 0, for which the reference tokenizer returns no ids (vits_tokenizer.cpp:200-208); pass ids");
        return none;
    }
    return process_ids_impl(model, ids.data(), ids.size());
    VITS_CATCH(none)
}

VITS_API vits_result vits_model_process_ids(vits_model* model, const int32_t* ids, size_t n_ids) {
    vits_result none{nullptr, 0};
    VITS_TRY
    return process_ids_impl(model, ids, n_ids);
    VITS_CATCH(none)
}

VITS_API int vits_model_set_mode(vits_model* model, int mode) {
    if (!model || (mode != VITS_MODE_REFERENCE && mode != VITS_MODE_HF)) {
        set_err("bad mode");
        return -1;
    }
    VITS_ENTER(model, -1)
    model->eng.mode = mode;
    return 0;
}
VITS_API int vits_model_get_mode(const vits_model* model) { return model ? model->eng.mode : -1; }
VITS_API void vits_reference_noise_seed(uint32_t seed) { vits::reference_noise_seed(seed); }
VITS_API int vits_model_set_arith(vits_model* model, int arith) {
    VITS_TRY
    if (!model || (arith != VITS_ARITH_F32 && arith != VITS_ARITH_BF16 && arith != VITS_ARITH_F16 && arith != VITS_ARITH_F32_SPLIT)) {
        set_err("bad arithmetic mode");
        return -1;
    }
    VITS_ENTER(model, -1)
    std::string err;
    if (model->eng.pending()) {
        set_err("batches in flight: call vits_model_wait for every submitted batch first");
        return -1;
    }
    if (model->eng.set_arith(arith, err) != 0) {
        set_err(err);
        return -1;
    }
    return 0;
    VITS_CATCH(-1)
}
VITS_API int vits_model_get_arith(const vits_model* model) { return model ? model->eng.arith : -1; }
VITS_API int vits_model_set_arith_scope(vits_model* model, int scope) {
    if (!model || (scope != VITS_ARITH_SCOPE_FLOW_VOCODER && scope != VITS_ARITH_SCOPE_ALL_CONVS)) {
        set_err("bad arithmetic scope");
        return -1;
    }
    VITS_ENTER(model, -1)
    if (model->eng.pending()) {
        set_err("batches in flight: call vits_model_wait for every submitted batch first");
        return -1;
    }
    model->eng.arith_scope = scope;
    return 0;
}
VITS_API int vits_model_get_arith_scope(const vits_model* model) { return model ? model->eng.arith_scope : -1; }
VITS_API int vits_model_set_ggml_tables(vits_model* model, int on) {
    VITS_TRY
    if (!model) {
        set_err("null argument");
        return -1;
    }
    VITS_ENTER(model, -1)
    if (model->eng.pending()) {
        set_err("batches in flight: call vits_model_wait for every submitted batch first");
        return -1;
    }
    std::string err;
    if (model->eng.set_ggml_tables(on, err) != 0) {
        set_err(err);
        return -1;
    }
    return 0;
    VITS_CATCH(-1)
}
VITS_API int vits_model_get_ggml_tables(const vits_model* model) { return model ? model->eng.ggml_tables : -1; }

VITS_API int vits_model_process_batch(vits_model* model, const int32_t* ids, const int32_t* id_lengths, int32_t batch, int32_t id_stride,
                                      const vits_process_opts* opts, vits_batch_result* out) {
    VITS_TRY
    if (out) std::memset(out, 0, sizeof(*out));
    if (!model || !ids) {
        set_err("null argument");
        return -1;
    }
    VITS_ENTER(model, -1)
    vits_process_opts o;
    std::memset(&o, 0, sizeof(o));
    o.mode = VITS_MODE_DEFAULT;
    o.noise_kind = VITS_NOISE_COUNTER;
    if (opts) std::memcpy(&o, opts, std::min<size_t>(sizeof(o), opts->struct_size ? opts->struct_size : sizeof(o)));
    std::string err;
    int rc;
    try {
        rc = model->eng.process_batch(ids, id_lengths, batch, id_stride, o, out, err);
    } catch (...) {
        if (out) vits_free_batch_result(out);  // (std::bad_alloc on the PCM buffer: nothing half-filled reaches the caller)
        throw;
    }
    if (rc != 0) {
        set_err(err);
        if (out) vits_free_batch_result(out);
    }
    return rc;
    VITS_CATCH(-1)
}

VITS_API int vits_model_submit_batch(vits_model* model, const int32_t* ids, const int32_t* id_lengths, int32_t batch, int32_t id_stride,
                                     const vits_process_opts* opts) {
    VITS_TRY
    if (!model || !ids) {
        set_err("null argument");
        return -1;
    }
    VITS_ENTER(model, -1)
    vits_process_opts o;
    std::memset(&o, 0, sizeof(o));
    o.mode = VITS_MODE_DEFAULT;
    o.noise_kind = VITS_NOISE_COUNTER;
    if (opts) std::memcpy(&o, opts, std::min<size_t>(sizeof(o), opts->struct_size ? opts->struct_size : sizeof(o)));
    std::string err;
    const int rc = model->eng.submit_batch(ids, id_lengths, batch, id_stride, o, err);
    if (rc != 0) set_err(err);
    return rc;
    VITS_CATCH(-1)
}

VITS_API int vits_model_wait(vits_model* model, vits_batch_result* out) {
    VITS_TRY
    if (out) std::memset(out, 0, sizeof(*out));
    if (!model) {
        set_err("null argument");
        return -1;
    }
    VITS_ENTER(model, -1)
    std::string err;
    int rc;
    try {
        rc = model->eng.wait_batch(out, err);
    } catch (...) {
        // (std::bad_alloc on the PCM buffer: the arrays already allocated go back, *out is left zeroed, the batch stays waitable)
        if (out) vits_free_batch_result(out);
        throw;
    }
    if (rc != 0) {
        set_err(err);
        if (out) vits_free_batch_result(out);
    }
    return rc;
    VITS_CATCH(-1)
}

VITS_API int vits_model_pending(const vits_model* model) { return model ? model->eng.pending() : -1; }

VITS_API void vits_free_batch_result(vits_batch_result* r) {
    if (!r) return;
    delete[] r->data;
    delete[] r->lengths;
    delete[] r->frames;
    std::memset(r, 0, sizeof(*r));
}

VITS_API int vits_model_sync(vits_model* model) {
    VITS_TRY
    if (!model) return -1;
    VITS_ENTER(model, -1)
    std::string err;
    const int rc = model->eng.sync(err);
    if (rc) set_err(err);
    return rc;
    VITS_CATCH(-1)
}

VITS_API int64_t vits_model_tokenize(vits_model* model, const char* text, int32_t* ids, size_t cap) {
    VITS_TRY
    if (!model || !text || (cap && !ids)) {
        set_err("null argument");
        return -1;
    }
    std::vector<int32_t> v;
    std::string terr;
    if (!model->eng.tok.tokenize_checked(text, v, terr)) {
        set_err(terr);
        return -1;
    }
    for (size_t i = 0; i < v.size() && i < cap; ++i) ids[i] = v[i];
    return (int64_t)v.size();
    VITS_CATCH(-1)
}

VITS_API int32_t vits_model_sampling_rate(const vits_model* model) { return model ? model->eng.hp.sampling_rate : 0; }
VITS_API int32_t vits_model_vocab_size(const vits_model* model) { return model ? model->eng.hp.vocab_size : 0; }
VITS_API int64_t vits_model_weight_bytes(const vits_model* model) { return model ? model->eng.weight_bytes : 0; }

VITS_API int64_t vits_model_get_tap(vits_model* model, const char* name, int32_t utt, float* dst, size_t cap) {
    VITS_TRY
    if (!model || !name) return 0;
    VITS_ENTER(model, 0)
    return model->eng.get_tap(name, utt, dst, cap);
    VITS_CATCH(0)
}

VITS_API int vits_synth_model_bytes(uint64_t seed, int32_t arch, char** bytes, size_t* size) {
    VITS_TRY
    if (!bytes || !size) return -1;
    vits::ModelFile f = vits::make_synthetic_model(seed, arch);
    std::vector<uint8_t> v = f.serialize();
    *bytes = new char[v.size()];
    std::memcpy(*bytes, v.data(), v.size());
    *size = v.size();
    return 0;
    VITS_CATCH(-1)
}
VITS_API void vits_free_bytes(char* bytes) { delete[] bytes; }

VITS_API int vits_model_file_reserialize(const char* in, size_t in_size, char** out, size_t* out_size) {
    VITS_TRY
    if (!in || !out || !out_size) return -1;
    vits::ModelFile f;
    std::string err;
    if (!f.parse(reinterpret_cast<const uint8_t*>(in), in_size, err)) {
        set_err(err);
        return -1;
    }
    std::vector<uint8_t> v = f.serialize();
    *out = new char[v.size()];
    std::memcpy(*out, v.data(), v.size());
    *out_size = v.size();
    return 0;
    VITS_CATCH(-1)
}

VITS_API int vits_model_file_validate(const char* bytes, size_t size) {
    VITS_TRY
    if (!bytes) {
        set_err("null argument");
        return -1;
    }
    vits::Engine e;
    std::string err;
    if (!e.validate(reinterpret_cast<const uint8_t*>(bytes), size, err)) {
        set_err(err);
        return -1;
    }
    return 0;
    VITS_CATCH(-1)
}

VITS_API int64_t vits_model_file_tokenize(const char* model_bytes, size_t size, const char* text, int32_t* ids, size_t cap) {
    VITS_TRY
    if (!model_bytes || !text) return -1;
    vits::ModelFile f;
    std::string err;
    if (!f.parse(reinterpret_cast<const uint8_t*>(model_bytes), size, err)) {
        set_err(err);
        return -1;
    }
    if (cap && !ids) {
        set_err("null argument");
        return -1;
    }
    vits::Tokenizer t;
    t.init(f);
    std::vector<int32_t> v;
    if (!t.tokenize_checked(text, v, err)) {
        set_err(err);
        return -1;
    }
    for (size_t i = 0; i < v.size() && i < cap; ++i) ids[i] = v[i];
    return (int64_t)v.size();
    VITS_CATCH(-1)
}

VITS_API int vits_prof_enable(vits_model* model, int32_t on) {
    if (!model) return -1;
    VITS_ENTER(model, -1)
    if (model->eng.pending()) {
        set_err("batches in flight: call vits_model_wait for every submitted batch first");
        return -1;
    }
    model->eng.prof.on = on != 0;
    return 0;
}
VITS_API int vits_prof_reset(vits_model* model) {
    if (!model) return -1;
    VITS_ENTER(model, -1)
    hipStreamSynchronize(model->eng.stream);
    model->eng.prof.reset();
    return 0;
}
VITS_API int64_t vits_prof_report(vits_model* model, char* buf, size_t cap) {
    VITS_TRY
    if (!model || !buf || !cap) return -1;
    VITS_ENTER(model, -1)
    hipStreamSynchronize(model->eng.stream);
    std::string s = model->eng.prof.report();
    const size_t n = std::min(cap - 1, s.size());
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
    return (int64_t)s.size();
    VITS_CATCH(-1)
}

// reference: test/main.cpp:31-33 (static_cast<short>(clamp(x, -1, 1) * 32767))
VITS_API void vits_pcm16_from_float(const float* pcm, size_t n, int16_t* out) {
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<int16_t>(std::max(-1.0f, std::min(1.0f, pcm[i])) * 32767);
}

VITS_API int vits_pcm16_from_float_device(const float* src, int64_t src_stride, int16_t* dst, int64_t dst_stride, const int64_t* lengths, int32_t rows,
                                          int64_t cols, void* hip_stream) {
    VITS_TRY
    if ((!src || !dst) && rows > 0 && cols > 0) {
        set_err("null argument");
        return -1;
    }
    const hipError_t e = vits::launch_pcm16(src, src_stride, dst, dst_stride, lengths, rows, cols, (hipStream_t)hip_stream);
    if (e != hipSuccess) {
        set_err(std::string("pcm16 kernel launch failed: ") + hipGetErrorString(e));
        return -1;
    }
    return 0;
    VITS_CATCH(-1)
}

// reference: test/main.cpp:23-63 (struct WAVHeader + write_wav)
VITS_API int vits_write_wav16(const char* path, const float* pcm, size_t n, int32_t sample_rate) {
    VITS_TRY
    if (!path || (!pcm && n)) return -1;
    std::vector<int16_t> s(n);
    vits_pcm16_from_float(pcm, n, s.data());
    const int32_t channels = 1, bits = 16;
    const int32_t data_bytes = (int32_t)(n * 2);
    unsigned char h[44];
    auto put32 = [&](int off, int32_t v) { std::memcpy(h + off, &v, 4); };
    auto put16 = [&](int off, int16_t v) { std::memcpy(h + off, &v, 2); };
    std::memcpy(h, "RIFF", 4);
    put32(4, 4 + (8 + 16) + (8 + data_bytes));
    std::memcpy(h + 8, "WAVE", 4);
    std::memcpy(h + 12, "fmt ", 4);
    put32(16, 16);
    put16(20, 1);
    put16(22, (int16_t)channels);
    put32(24, sample_rate);
    put32(28, sample_rate * channels * (bits / 8));
    put16(32, (int16_t)(channels * (bits / 8)));
    put16(34, (int16_t)bits);
    std::memcpy(h + 36, "data", 4);
    put32(40, data_bytes);
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) {
        set_err(std::string("cannot open ") + path);
        return -1;
    }
    f.write(reinterpret_cast<const char*>(h), 44);
    f.write(reinterpret_cast<const char*>(s.data()), data_bytes);
    return f.good() ? 0 : -1;
    VITS_CATCH(-1)
}

VITS_API int vits_set_device(int32_t device) {
    if (hipSetDevice(device) != hipSuccess) {
        set_err("hipSetDevice failed");
        return -1;
    }
    return 0;
}

VITS_API int vits_device_info(char* name, size_t cap, int32_t* cu_count, int32_t* clock_mhz, int64_t* hbm_bytes) {
    VITS_TRY
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_err("no HIP device");
        return -1;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
        set_err("hipGetDeviceProperties failed");
        return -1;
    }
    if (name && cap) {
        std::snprintf(name, cap, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (clock_mhz) *clock_mhz = p.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return 0;
    VITS_CATCH(-1)
}

// ---- operator-level entry points: host in, host out ---------------------------------------------------------
namespace {
struct DevBuf {
    float* p = nullptr;
    ~DevBuf() {
        if (p) hipFree(p);
    }
    bool put(const float* h, size_t n) {
        if (hipMalloc((void**)&p, std::max<size_t>(n, 1) * 4) != hipSuccess) return false;
        if (h) return hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice) == hipSuccess;
        return hipMemset(p, 0, n * 4) == hipSuccess;
    }
};
struct DevInts {
    int* p = nullptr;
    ~DevInts() {
        if (p) hipFree(p);
    }
    bool put(const int32_t* h, size_t n) {
        if (!h) return true;
        if (hipMalloc((void**)&p, n * 4) != hipSuccess) return false;
        return hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
};
vits::TensorRef tref(float* p, int channels, int stride) {
    vits::TensorRef t;
    t.p = p;
    t.cs = stride;
    t.bs = (int64_t)channels * stride;
    return t;
}
int fail(const char* what) {
    set_err(what);
    return -1;
}
thread_local int g_op_arith = VITS_ARITH_F32;
// one operator-level conv in the selected arithmetic: the fp32 MFMA kernel, or (16-bit modes) the converter + conv16 pair the
// engine uses for fp32-layout tensors
hipError_t run_op_conv(vits::PackedConv& pc, const vits::ConvCall& c, const float* w_torch, int k, DevBuf& w16, DevBuf& x16) {
    using namespace vits;
    if (g_op_arith == VITS_ARITH_F32) return launch_conv(pc, c, nullptr);
    const std::vector<uint16_t> packed = pack_conv_weights16(w_torch, pc.cout, pc.cin, k, pc.epi, pc.ct_stride, g_op_arith);
    if (!w16.put(nullptr, packed.size() / 2 + 1)) return hipErrorOutOfMemory;
    if (hipMemcpy(w16.p, packed.data(), packed.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return hipErrorUnknown;
    pc.wp16 = reinterpret_cast<uint16_t*>(w16.p);
    Ref16 r;
    r.ts = (c.t_in + 7) / 8 * 8;
    r.bs = (int64_t)((pc.cin + 7) / 8) * r.ts * 8;
    if (!x16.put(nullptr, (size_t)c.batch * r.bs / 2 + 1)) return hipErrorOutOfMemory;
    r.p = reinterpret_cast<uint16_t*>(x16.p);
    hipError_t e = launch_to_group16(c.x, c.len_in, c.batch, pc.cin, c.t_in, c.pre_act ? c.slope : 1.0f, r, g_op_arith, nullptr);
    if (e != hipSuccess) return e;
    Conv16Call q;
    q.x = r;
    q.len_in = c.len_in;
    q.len_out = c.len_out;
    q.batch = c.batch;
    q.t_in = c.t_in;
    q.t_out = c.t_out;
    q.dil = c.dil;
    q.pad_l = c.pad_l;
    q.post_act = c.post_act;
    q.post_slope = c.post_slope;
    q.scale = c.scale;
    q.scale_div = c.scale_div;
    q.ct_crop = c.ct_crop;
    q.y = c.y;
    q.res = c.res;
    q.acc = c.acc;
    q.tile = c.tile;
    return launch_conv16(pc, q, g_op_arith, nullptr);
}
}  // namespace

VITS_API int vits_op_set_arith(int32_t arith) {
    if (arith != VITS_ARITH_F32 && arith != VITS_ARITH_BF16 && arith != VITS_ARITH_F16) return fail("bad arithmetic mode");
    g_op_arith = arith;
    return 0;
}

VITS_API int vits_op_conv1d(const vits_conv1d_desc* d, const float* x, const float* w, const float* bias, const float* residual, const float* accum,
                            const int32_t* lens, float* y) {
    VITS_TRY
    using namespace vits;
    if (!d || !x || !w || !y) return fail("null argument");
    const int gate = d->post_act == 2;
    const int cy = gate ? d->cout / 2 : d->cout;
    PackedConv pc;
    pc.cin = d->cin;
    pc.cout = d->cout;
    pc.kt = d->k;
    pc.epi = gate ? EPI_GATE : EPI_STD;
    std::vector<float> packed = pack_conv_weights(w, d->cout, d->cin, d->k, pc.epi, 0, &pc.rows, &pc.mtiles_used, &pc.mtiles, &pc.nchunks);
    DevBuf dw, dwl, db, dx, dy, dr, da;
    DevInts dl;
    const size_t nx = (size_t)d->batch * d->cin * d->t_stride, ny = (size_t)d->batch * cy * d->t_stride;
    if (conv_lat16_candidate(pc.epi, d->k, d->cin)) {
        const std::vector<float> pl = repack_conv_weights_l16(packed, pc.mtiles, pc.nchunks, d->k);
        if (!dwl.put(pl.data(), pl.size())) return fail("device allocation failed");
        pc.wp_l16 = dwl.p;
    }
    if (!dw.put(packed.data(), packed.size()) || !dx.put(x, nx) || !dy.put(nullptr, ny) || !dl.put(lens, d->batch)) return fail("device allocation failed");
    if (bias && !db.put(bias, d->cout)) return fail("device allocation failed");
    if (residual && !dr.put(residual, ny)) return fail("device allocation failed");
    if (accum && !da.put(accum, ny)) return fail("device allocation failed");
    pc.wp = dw.p;
    pc.bias = db.p;
    ConvCall c;
    c.x = tref(dx.p, d->cin, d->t_stride);
    c.y = tref(dy.p, cy, d->t_stride);
    if (residual) c.res = tref(dr.p, cy, d->t_stride);
    if (accum) c.acc = tref(da.p, cy, d->t_stride);
    c.len_in = dl.p;
    c.len_out = dl.p;
    c.batch = d->batch;
    c.t_in = c.t_out = d->t;
    c.dil = d->dilation;
    c.pad_l = d->pad_left;
    c.pre_act = d->pre_act;
    c.slope = d->pre_slope;
    c.post_act = d->post_act == 1 ? 1 : 0;
    c.scale = d->out_scale;
    DevBuf w16, x16;
    hipError_t e = run_op_conv(pc, c, w, d->k, w16, x16);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(hipGetErrorString(e));
    if (hipMemcpy(y, dy.p, ny * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail("copy back failed");
    return 0;
    VITS_CATCH(-1)
}

VITS_API int vits_op_conv_transpose1d(const vits_convt1d_desc* d, const float* x, const float* w, const float* bias, const int32_t* lens, float* y) {
    VITS_TRY
    using namespace vits;
    if (!d || !x || !w || !y) return fail("null argument");
    if (d->k != 2 * d->stride) return fail("kernel must be 2*stride");
    PackedConv pc;
    pc.cin = d->cin;
    pc.cout = d->cout;
    pc.kt = 2;
    pc.epi = EPI_CONVT;
    pc.ct_stride = d->stride;
    std::vector<float> packed = pack_conv_weights(w, d->cout, d->cin, d->k, EPI_CONVT, d->stride, &pc.rows, &pc.mtiles_used, &pc.mtiles, &pc.nchunks);
    DevBuf dw, db, dx, dy;
    DevInts dl, dlo;
    const size_t nx = (size_t)d->batch * d->cin * d->t_stride, ny = (size_t)d->batch * d->cout * d->t_out_stride;
    std::vector<int32_t> lo;
    if (lens) {
        lo.resize(d->batch);
        for (int b = 0; b < d->batch; ++b) lo[b] = d->stride * lens[b] + d->k - d->stride - 2 * d->crop;
    }
    if (!dw.put(packed.data(), packed.size()) || !dx.put(x, nx) || !dy.put(nullptr, ny) || !dl.put(lens, d->batch) || !dlo.put(lens ? lo.data() : nullptr, d->batch))
        return fail("device allocation failed");
    if (bias && !db.put(bias, d->cout)) return fail("device allocation failed");
    pc.wp = dw.p;
    pc.bias = db.p;
    ConvCall c;
    c.x = tref(dx.p, d->cin, d->t_stride);
    c.y = tref(dy.p, d->cout, d->t_out_stride);
    c.len_in = dl.p;
    c.len_out = dlo.p;
    c.batch = d->batch;
    c.t_in = d->t;
    c.t_out = d->stride * d->t + d->k - d->stride - 2 * d->crop;
    c.pre_act = d->pre_slope != 1.0f;
    c.slope = d->pre_slope;
    c.ct_crop = d->crop;
    DevBuf w16, x16;
    hipError_t e = run_op_conv(pc, c, w, d->k, w16, x16);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(hipGetErrorString(e));
    if (hipMemcpy(y, dy.p, ny * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail("copy back failed");
    return 0;
    VITS_CATCH(-1)
}

VITS_API int vits_op_rel_attention(int32_t batch, int32_t heads, int32_t head_dim, int32_t t, int32_t t_stride, int32_t window, const float* q, const float* k,
                                   const float* v, const float* rel_k, const float* rel_v, const int32_t* lens, float* out) {
    VITS_TRY
    using namespace vits;
    const int C = heads * head_dim;
    const size_t n = (size_t)batch * C * t_stride, nr = (size_t)(2 * window + 1) * head_dim;
    DevBuf dq, dk, dv, drk, drv, dout;
    DevInts dl;
    if (!dq.put(q, n) || !dk.put(k, n) || !dv.put(v, n) || !drk.put(rel_k, nr) || !drv.put(rel_v, nr) || !dout.put(nullptr, n) || !dl.put(lens, batch))
        return fail("device allocation failed");
    hipError_t e = launch_rel_attention(tref(dq.p, C, t_stride), tref(dk.p, C, t_stride), tref(dv.p, C, t_stride), drk.p, drv.p, tref(dout.p, C, t_stride), dl.p, batch,
                                        heads, head_dim, t, window, 1.0f, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(hipGetErrorString(e));
    if (hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail("copy back failed");
    return 0;
    VITS_CATCH(-1)
}

VITS_API int vits_op_add_layer_norm(int32_t batch, int32_t channels, int32_t t, int32_t t_stride, float eps, const float* x, const float* residual,
                                    const float* gamma, const float* beta, float* y) {
    VITS_TRY
    using namespace vits;
    const size_t n = (size_t)batch * channels * t_stride;
    DevBuf dx, dr, dg, db, dy;
    if (!dx.put(x, n) || !dg.put(gamma, channels) || !db.put(beta, channels) || !dy.put(nullptr, n)) return fail("device allocation failed");
    if (residual && !dr.put(residual, n)) return fail("device allocation failed");
    TensorRef none;
    hipError_t e = launch_add_layer_norm(tref(dx.p, channels, t_stride), residual ? tref(dr.p, channels, t_stride) : none, dg.p, db.p, tref(dy.p, channels, t_stride),
                                         nullptr, batch, channels, t, eps, 0, none, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(hipGetErrorString(e));
    if (hipMemcpy(y, dy.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail("copy back failed");
    return 0;
    VITS_CATCH(-1)
}

// ---- multi-GPU PCM gather (pcm_gather.cpp) ---------------------------------------------------------------------------------------
struct vits_gather_ctx {
    vits::PcmGather g;
    std::atomic<bool> busy{false};
};

VITS_API int vits_pcm_gather_unique_id(char* id_out) {
    VITS_TRY
    if (!id_out) {
        set_err("null argument");
        return -1;
    }
    const vits::RcclApi& api = vits::RcclApi::get();
    if (!api.ok()) {
        set_err(api.why);
        return -1;
    }
    vits::RcclApi::UniqueId uid;
    std::memset(&uid, 0, sizeof(uid));
    const int r = api.GetUniqueId(&uid);
    if (r != 0) {
        set_err(std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
        return -1;
    }
    std::memcpy(id_out, uid.internal, sizeof(uid.internal));
    return 0;
    VITS_CATCH(-1)
}

VITS_API vits_gather_ctx* vits_pcm_gather_init(const char* id, size_t id_bytes, int32_t rank, int32_t world, int32_t rows, int64_t row_capacity, int32_t elem_bytes) {
    VITS_TRY
    vits_gather_ctx* h = new vits_gather_ctx();
    std::string err;
    if (!h->g.init(id, id_bytes, rank, world, rows, row_capacity, elem_bytes, err)) {
        set_err(err);
        delete h;
        return nullptr;
    }
    return h;
    VITS_CATCH(nullptr)
}

VITS_API int vits_pcm_gather(vits_gather_ctx* g, const void* pcm_device, int64_t pcm_stride, const int64_t* lengths_host, void* hip_stream, vits_gather_result* out) {
    VITS_TRY
    if (!g || !out) {  // (nothing to hand a result to, or no communicator to tell the peers through)
        set_err("null argument");
        return -1;
    }
    std::memset(out, 0, sizeof(*out));
    vits::BusyGuard guard(&g->busy);
    if (!guard.entered()) {
        set_err("gather busy: one call at a time per gather object");
        return -1;
    }
    vits::PcmGather::Result r;
    std::string err;
    if (g->g.gather(pcm_device, pcm_stride, lengths_host, (hipStream_t)hip_stream, &r, err) != 0) {
        set_err(err);
        return -1;
    }
    out->data = r.data;
    out->stride = r.stride;
    out->lengths = r.lengths;
    out->rows_total = r.rows_total;
    return 0;
    VITS_CATCH(-1)
}

VITS_API int vits_pcm_gather_verdict(const int64_t* table, int32_t world, int32_t rows, int64_t* stride_out) {
    VITS_TRY
    if (!table || !stride_out || world < 1 || rows < 1) {
        set_err("null argument");
        return -1;
    }
    std::string err;
    if (vits::PcmGather::verdict(table, world, rows, stride_out, err) != 0) {
        set_err(err);
        return -1;
    }
    return 0;
    VITS_CATCH(-1)
}

VITS_API void vits_pcm_gather_destroy(vits_gather_ctx* g) {
    try {
        delete g;
    } catch (...) {
    }
}
