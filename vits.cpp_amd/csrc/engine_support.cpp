// engine_support.cpp — the engine's plumbing: roctx ranges, the reference noise stream, tokenizer, HIP-event profiler, arenas, knobs.
#include <dlfcn.h>

#include <atomic>
#include <mutex>
#include <random>
#include <sstream>
#include <thread>

#include "engine_internal.h"

#include <condition_variable>

namespace vits {

RoctxApi::RoctxApi() {
    if (!std::getenv("VITS_ROCTX")) return;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
}
const RoctxApi& roctx_api() {
    static const RoctxApi api;
    return api;
}

void Knobs::read() {
    auto flag = [](const char* name) { return std::getenv(name) != nullptr; };
    if (const char* e = std::getenv("VITS_RB_STREAMS")) rb_streams = std::atoi(e) >= 2 ? 3 : 1;
    if (const char* e = std::getenv("VITS_LRELU_COPY_MINC")) lrelu_copy_minc = std::atoi(e);
    if (const char* e = std::getenv("VITS_RB16_SERIAL_MAX_FRAMES")) rb16_serial_max_frames = std::atoi(e);
    if (const char* e = std::getenv("VITS_RB16_SERIAL_MIN_FRAMES")) rb16_serial_min_frames = std::atoi(e);
    if (const char* e = std::getenv("VITS_RB32_SUM3_MAX_FRAMES")) rb32_sum3_max_frames = std::atoi(e);
    if (const char* e = std::getenv("VITS_LAT16_LAZY_TOKENS")) lat16_lazy_tokens = std::atoi(e);
    lat16_eager = flag("VITS_LAT16_EAGER");
    no_dds_fuse = flag("VITS_NO_DDS_FUSE");
    no_wn_fuse = flag("VITS_NO_WN_FUSE");
    no_group16 = flag("VITS_NO_GROUP16");
    no_fuse16 = flag("VITS_NO_FUSE16");
    no_rbblock16 = flag("VITS_NO_RBBLOCK16");
    no_fuse32 = flag("VITS_NO_FUSE32");
    no_rbblock32 = flag("VITS_NO_RBBLOCK32");
    no_rb_group = flag("VITS_NO_RB_GROUP");
    rb_group_always = flag("VITS_RB_GROUP");
    no_flow_fuse = flag("VITS_NO_FLOW_FUSE");
    if (const char* e = std::getenv("VITS_PROF_ATTACH")) prof_attach = std::atoi(e) != 0;
    if (const char* e = std::getenv("VITS_FRONT_PRIO")) front_prio = std::atoi(e);
    no_pipeline = flag("VITS_NO_PIPELINE");
    if (const char* e = std::getenv("VITS_REF_AHEAD_FRAMES_PER_ID")) ref_ahead_frames_per_id = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("VITS_FLOW_CHAINS")) flow_chains = std::atoi(e);
    if (const char* e = std::getenv("VITS_FLOW_CHAIN_MIN_BLOCKS")) flow_chain_min_blocks = std::atoi(e);
    if (const char* e = std::getenv("VITS_SPLIT_MIN_BATCH")) split_min_batch = std::atoi(e);
    if (const char* e = std::getenv("VITS_SPLIT_FIRST_PCT")) split_first_pct = std::min(95, std::max(5, std::atoi(e)));
    keep_stage_sum32 = flag("VITS_KEEP_STAGE_SUM32");
    kernel = KernelKnobs::from_env();
}

// ---- reference noise stream (vits.cpp:31 global engine; ggml-util.h:187-199 fresh distribution per tensor) ----
static std::default_random_engine g_ref_rng;
static std::mutex g_ref_mu;
void reference_noise_seed(uint32_t seed) {
    std::lock_guard<std::mutex> lk(g_ref_mu);
    g_ref_rng.seed(seed);
}
void reference_noise_fill(float* dst, size_t n) {
    std::lock_guard<std::mutex> lk(g_ref_mu);
    std::normal_distribution<float> dist(0.0f, 1.0f);
    for (size_t i = 0; i < n; ++i) dst[i] = dist(g_ref_rng);
}

// ---- the reference stream of ONE batch-1 call, drawn on a helper thread while the device runs stage one -----------------------------------
// vits.cpp draws tensor_randn {T, 2} at :948 and tensor_randn_like(prior_means) = [L, 192] at :1059, the second only once L is known, i.e. behind
// stage one; on the host that is ~1.25 ms for 128 ids (libstdc++'s polar method, ~28 ns per value) during which the device waits — while during stage
// one the calling thread is busy queueing ~70 launches (at batch 1 the host barely keeps ahead of the device). The values are one sequential stream
// whatever L turns out to be (element n of the tensor is draw n), so a helper thread draws them from the start of the call; the engine state after every
// value is kept, and finish(n) leaves the global engine exactly where n draws leave it (a fresh normal_distribution per tensor, ggml-util.h:187-199: a
// cached second value of the polar method is discarded with the distribution object). The stream's lock is held from start to finish.
struct RefNoiseAhead::Impl {
    // ONE helper thread per engine, created at the first reference-noise call and parked on a condition variable between calls (round 5 spawned a
    // std::thread per call and let it spin on yield() at the buffer's end)
    std::thread th;
    std::mutex m;
    std::condition_variable cv;        // worker: a job, a new target / capacity, or quit
    std::condition_variable cv_done;   // caller: the job has ended
    bool have_job = false, quit = false;
    std::atomic<bool> done{true};
    std::vector<std::default_random_engine> states;  // states[i] = engine after prior value i + 1
    std::default_random_engine start;                // ... and before the first one
    std::vector<float> dur;
    std::atomic<int> dur_ready{0};
    std::atomic<size_t> target{SIZE_MAX}, drawn{0}, cap{0};
    std::atomic<float*> buf{nullptr};
    void job() {
        std::lock_guard<std::mutex> lk(g_ref_mu);  // the process-global stream (vits.cpp:31) belongs to this call until finish()
        {
            std::normal_distribution<float> dist(0.0f, 1.0f);  // vits.cpp:948
            for (float& v : dur) v = dist(g_ref_rng);
        }
        dur_ready.store(1, std::memory_order_release);
        std::normal_distribution<float> dist(0.0f, 1.0f);  // vits.cpp:1059
        start = g_ref_rng;
        size_t n = 0;
        for (;;) {
            const size_t tgt = target.load(std::memory_order_acquire);
            const size_t lim = std::min(tgt, cap.load(std::memory_order_acquire));
            if (n < lim) {
                float* b = buf.load(std::memory_order_acquire);
                const size_t end = std::min(lim, n + 256);
                for (; n < end; ++n) {
                    b[n] = dist(g_ref_rng);
                    states[n] = g_ref_rng;
                }
                drawn.store(n, std::memory_order_release);
                continue;
            }
            if (tgt != SIZE_MAX && n >= tgt) {
                if (n > tgt) g_ref_rng = tgt > 0 ? states[tgt - 1] : start;
                return;
            }
            // at the capacity with the size still unknown (or a larger buffer on its way): sleep until finish() / rebase() says more
            std::unique_lock<std::mutex> ul(m);
            cv.wait(ul, [&] { return target.load(std::memory_order_acquire) != tgt || cap.load(std::memory_order_acquire) > n; });
        }
    }
    void run() {
        for (;;) {
            {
                std::unique_lock<std::mutex> ul(m);
                cv.wait(ul, [&] { return have_job || quit; });
                if (quit) return;
                have_job = false;
            }
            job();
            {
                std::lock_guard<std::mutex> lk(m);
                done.store(true, std::memory_order_release);
            }
            cv_done.notify_all();
        }
    }
};
RefNoiseAhead::RefNoiseAhead() : impl_(new Impl) {}
RefNoiseAhead::~RefNoiseAhead() {
    if (active_) finish(0);
    if (impl_->th.joinable()) {
        {
            std::lock_guard<std::mutex> lk(impl_->m);
            impl_->quit = true;
        }
        impl_->cv.notify_all();
        impl_->th.join();
    }
    delete impl_;
}
void RefNoiseAhead::start(size_t n_dur, float* prior, size_t cap) {
    Impl& I = *impl_;
    I.dur.assign(n_dur, 0.f);
    I.dur_ready.store(0);
    I.target.store(SIZE_MAX);
    I.drawn.store(0);
    if (I.states.size() < cap) I.states.resize(cap);
    I.cap.store(cap);
    I.buf.store(prior);
    if (!I.th.joinable()) I.th = std::thread([this] { impl_->run(); });
    {
        std::lock_guard<std::mutex> lk(I.m);
        I.done.store(false, std::memory_order_release);
        I.have_job = true;
    }
    I.cv.notify_all();
    active_ = true;
}
const float* RefNoiseAhead::duration_noise() {
    while (!impl_->dur_ready.load(std::memory_order_acquire)) std::this_thread::yield();  // (256 draws, ~10 us after start(): asked for a whole text encoder later)
    return impl_->dur.data();
}
size_t RefNoiseAhead::drawn() const { return impl_->drawn.load(std::memory_order_acquire); }
size_t RefNoiseAhead::capacity() const { return impl_->cap.load(std::memory_order_acquire); }
void RefNoiseAhead::rebase(float* prior, size_t cap) {
    // only while the worker rests at the old capacity (drawn() == capacity()): nothing of its state is being touched
    Impl& I = *impl_;
    if (I.states.size() < cap) I.states.resize(cap);
    {
        std::lock_guard<std::mutex> lk(I.m);
        I.buf.store(prior, std::memory_order_release);
        I.cap.store(cap, std::memory_order_release);
    }
    I.cv.notify_all();
}
void RefNoiseAhead::finish(size_t n) {
    if (!active_) return;
    Impl& I = *impl_;
    {
        std::lock_guard<std::mutex> lk(I.m);
        I.target.store(n, std::memory_order_release);
    }
    I.cv.notify_all();
    // the worker is usually past n already (or a few hundred draws short): look before sleeping
    for (int spin = 0; spin < 2000 && !I.done.load(std::memory_order_acquire); ++spin) std::this_thread::yield();
    if (!I.done.load(std::memory_order_acquire)) {
        std::unique_lock<std::mutex> ul(I.m);
        I.cv_done.wait(ul, [&] { return I.done.load(std::memory_order_acquire); });
    }
    active_ = false;
}

// ---- tokenizer (src/vits_tokenizer.cpp:57-78,182-208; deterministic longest match instead of unordered_map order, Q11) ----
void Tokenizer::init(const ModelFile& f) {
    vocab.clear();
    for (auto& kv : f.vocab) vocab.emplace_back(kv.first, (int32_t)kv.second);
    std::stable_sort(vocab.begin(), vocab.end(), [](auto& a, auto& b) { return a.first.size() > b.first.size(); });
    add_blank = f.add_blank != 0;
    blank_id = 0;
    for (auto& kv : f.vocab)
        if (kv.first == f.pad_token) blank_id = (int32_t)kv.second;  // vocab[pad_token], vits_tokenizer.cpp:201
    phonetic = false;
    for (auto& kv : f.config)
        if (kv.first == "phonetic" && kv.second == "1") phonetic = true;  // vits_model_data.cpp:92-94
}

bool Tokenizer::tokenize_checked(const std::string& text, std::vector<int32_t>& ids, std::string& err) const {
    if (phonetic) {
        err = "model expects espeak phonemes (config phonetic=1; the reference needs VITS_ESPEAK for it, vits_tokenizer.cpp:176-178): pass ids "
              "(vits_model_process_ids / vits_model_process_batch)";
        return false;
    }
    ids = tokenize(text);
    return true;
}

std::vector<int32_t> Tokenizer::tokenize(const std::string& text) const {
    std::string s = text;
    for (auto& c : s) c = (char)std::tolower((unsigned char)c);  // :195-197
    std::vector<int32_t> toks;
    size_t i = 0;
    while (i < s.size()) {
        bool found = false;
        for (auto& kv : vocab) {
            if (!kv.first.empty() && s.compare(i, kv.first.size(), kv.first) == 0) {
                toks.push_back(kv.second);
                i += kv.first.size();
                found = true;
                break;
            }
        }
        if (!found) i++;  // unknown bytes are skipped (:72-75)
    }
    std::vector<int32_t> fin;
    if (add_blank) {  // :200-206 ; without add_blank the reference returns an empty vector
        fin.assign(toks.size() * 2 + 1, blank_id);
        for (size_t k = 0; k < toks.size(); ++k) fin[k * 2 + 1] = toks[k];
    }
    return fin;
}

// ---- profiler -----------------------------------------------------------------------------------------
hipEvent_t Profiler::get() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void Profiler::begin(const char* name, double flop, double bytes, hipStream_t s, bool chain) {
    if (!on) return;
    auto it = ids.find(name);
    int id;
    if (it == ids.end()) {
        id = (int)names.size();
        names.push_back(name);
        ids[name] = id;
        agg.emplace_back();
    } else
        id = it->second;
    if (attach) {
        Rec r{id, get(), get(), flop, bytes, false};
        recs.push_back(r);
        vits_launch_timer = LaunchTimer{r.a, r.b, 0};  // the next launch of this thread takes them along
        last_ok = false;
        return;
    }
    const bool share = chain && last_ok && last_s == s;
    Rec r{id, share ? last_b : get(), get(), flop, bytes, share};
    if (!share) hipEventRecord(r.a, s);
    recs.push_back(r);
    last_ok = false;
}
void Profiler::end(hipStream_t s) {
    if (!on || recs.empty()) return;
    if (attach) {
        LaunchTimer& lt = vits_launch_timer;
        Rec& r = recs.back();
        if (lt.start == r.a && lt.launches != 1) {
            // several launches in one span: the start is the first kernel's, the stop a recorded event behind the last one (the attached stop
            // event belongs to the first dispatch). No launch at all: both recorded here (an empty span).
            if (lt.launches == 0) hipEventRecord(r.a, s);
            r.spare = r.b;
            r.b = get();
            hipEventRecord(r.b, s);
        }
        lt = LaunchTimer{};
        return;
    }
    hipEventRecord(recs.back().b, s);
    last_b = recs.back().b;
    last_s = s;
    last_ok = true;
}
void Profiler::collect() {
    for (auto& r : recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            Agg& a = agg[r.name_id];
            a.calls++;
            a.ms += ms;
            a.flop += r.flop;
            a.bytes += r.bytes;
        }
        if (!r.a_shared) pool.push_back(r.a);
        pool.push_back(r.b);
        if (r.spare) pool.push_back(r.spare);
    }
    recs.clear();
    last_ok = false;
}
void Profiler::reset() {
    collect();
    for (auto& a : agg) a = Agg();
}
std::string Profiler::report() {
    collect();
    std::ostringstream o;
    o.precision(9);
    o << "{\"kernels\":[";
    bool first = true;
    for (size_t i = 0; i < names.size(); ++i) {
        if (!agg[i].calls) continue;
        o << (first ? "" : ",") << "{\"name\":\"" << names[i] << "\",\"calls\":" << agg[i].calls << ",\"ms\":" << agg[i].ms << ",\"flop\":" << agg[i].flop
          << ",\"bytes\":" << agg[i].bytes << "}";
        first = false;
    }
    o << "]}";
    return o.str();
}
Profiler::~Profiler() {
    collect();
    for (auto e : pool) hipEventDestroy(e);
}

// ---- arena ----------------------------------------------------------------------------------------------
hipError_t Arena::reserve(size_t bytes) {
    off = 0;
    if (bytes <= cap) return hipSuccess;
    if (base) hipFree(base);
    base = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + (1 << 20);
    hipError_t e = hipMalloc((void**)&base, want);
    if (e == hipSuccess) cap = want;
    return e;
}
Arena::~Arena() {
    if (base) hipFree(base);
}


}  // namespace vits
