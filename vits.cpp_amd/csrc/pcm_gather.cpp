// pcm_gather.cpp — see pcm_gather.h. Host C++ only (no kernels): copies, two RCCL all-gathers, one stream.
#include "pcm_gather.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace vits {

const RcclApi& RcclApi::get() {
    static const RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;
        if (const char* e = std::getenv("VITS_RCCL_LIB")) {
            h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
            if (!h) a.why = std::string("dlopen(VITS_RCCL_LIB=") + e + ") failed: " + dlerror();
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (!h) a.why = "librccl.so not found (dlopen): multi-GPU gather needs RCCL";
        }
        if (!h) return a;
        auto sym = [&](const char* n) { return dlsym(h, n); };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        auto ag = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GetErrorString || !ag) {
            a.why = "the RCCL library lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclGetErrorString / ncclAllGather";
            return a;
        }
        a.AllGather = ag;
        a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(sym("ncclCommAbort"));  // optional: only the failure path wants it
        return a;
    }();
    return api;
}

#define G_HIP(call)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return fail;                                                                     \
        }                                                                                    \
    } while (0)
#define G_NCCL(call)                                                                         \
    do {                                                                                     \
        int r_ = (call);                                                                     \
        if (r_ != 0) {                                                                       \
            err = std::string(#call) + ": " + RcclApi::get().GetErrorString(r_);             \
            return fail;                                                                     \
        }                                                                                    \
    } while (0)

PcmGather::~PcmGather() {
    if (side_) hipStreamSynchronize(side_);
    if (comm_) RcclApi::get().CommDestroy(comm_);  // (an aborted communicator was already given back by poison())
    if (len_send_h_) hipHostFree(len_send_h_);
    if (len_all_h_) hipHostFree(len_all_h_);
    for (void* p : {(void*)len_send_d_, (void*)len_all_d_, (void*)send_, (void*)out_})
        if (p) hipFree(p);
    if (ev_) hipEventDestroy(ev_);
    if (side_) hipStreamDestroy(side_);
}

bool PcmGather::init(const char* id, size_t id_bytes, int rank, int world, int rows, int64_t capacity, int elem_bytes, std::string& err) {
    const bool fail = false;
    if (world < 1 || rank < 0 || rank >= world || rows < 1 || capacity < 1 || (elem_bytes != 4 && elem_bytes != 2)) {
        err = "vits_pcm_gather_init: need 0 <= rank < world, rows >= 1, row_capacity >= 1, elem_bytes 4 (fp32) or 2 (pcm16)";
        return false;
    }
    rank_ = rank, world_ = world, rows_ = rows, cap_ = capacity, eb_ = elem_bytes;
    // world == 1 needs no communicator (VITS_GATHER_FORCE_RCCL=1 makes one anyway: the RCCL calls of the N > 1 path on a 1-GPU box)
    if (world > 1 || std::getenv("VITS_GATHER_FORCE_RCCL")) {
        const RcclApi& api = RcclApi::get();
        if (!api.ok()) {
            err = api.why;
            return false;
        }
        if (!id || id_bytes != sizeof(RcclApi::UniqueId)) {
            err = "vits_pcm_gather_init: unique id must be the 128 bytes vits_pcm_gather_unique_id wrote on rank 0";
            return false;
        }
        RcclApi::UniqueId uid;
        std::memcpy(uid.internal, id, sizeof(uid.internal));
        G_NCCL(api.CommInitRank(&comm_, world, uid, rank));
    }
    const size_t n = (size_t)world * rows;
    G_HIP(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
    G_HIP(hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
    // the first all-gather carries [row_capacity, lengths...] per rank (see verdict())
    const size_t tab = (size_t)rows + 1;
    len_out_h_.assign(n, 0);
    G_HIP(hipHostMalloc((void**)&len_send_h_, sizeof(int64_t) * tab, hipHostMallocDefault));
    G_HIP(hipHostMalloc((void**)&len_all_h_, sizeof(int64_t) * tab * world, hipHostMallocDefault));
    G_HIP(hipMalloc((void**)&len_send_d_, sizeof(int64_t) * tab));
    G_HIP(hipMalloc((void**)&len_all_d_, sizeof(int64_t) * tab * world));
    G_HIP(hipMalloc((void**)&send_, (size_t)rows * capacity * eb_));
    G_HIP(hipMalloc((void**)&out_, n * (size_t)capacity * eb_));
    return true;
}

// The decision every rank must reach TOGETHER. table = what the first all-gather delivered: per rank [row_capacity, length of row 0, ...]; a
// rank that failed its own checks (null buffer, a row longer than its pcm_stride or its capacity, a negative length) sent -1 for the row. The
// verdict is a pure function of the gathered table, which is the same on every rank, so either all ranks enter the second all-gather or none does:
// a rank-local early return here left the peers blocked in ncclAllGather for good.
int PcmGather::verdict(const int64_t* table, int world, int rows, int64_t* smax_out, std::string& err) {
    int64_t cap = table[0], smax = 1;
    for (int r = 0; r < world; ++r) {
        const int64_t* t = table + (size_t)r * (rows + 1);
        if (t[0] != cap) {
            err = "vits_pcm_gather: rank " + std::to_string(r) + " was initialised with row_capacity " + std::to_string(t[0]) + ", rank 0 with " + std::to_string(cap) +
                  " (all ranks must agree)";
            return -1;
        }
    }
    for (int r = 0; r < world; ++r) {
        const int64_t* t = table + (size_t)r * (rows + 1);
        for (int b = 0; b < rows; ++b) {
            if (t[1 + b] < 0) {
                err = "vits_pcm_gather: rank " + std::to_string(r) + " passed an unusable row " + std::to_string(b) +
                      " (null buffer, negative length, or a row longer than that rank's pcm_stride / the row_capacity agreed at init); no rank exchanged PCM";
                return -1;
            }
            if (t[1 + b] > cap) {
                err = "vits_pcm_gather: row " + std::to_string(b) + " of rank " + std::to_string(r) + " is longer than row_capacity";
                return -1;
            }
            smax = std::max(smax, t[1 + b]);
        }
    }
    *smax_out = smax;
    return 0;
}

void PcmGather::poison() {
    broken_ = true;
    if (comm_) {
        const RcclApi& api = RcclApi::get();
        if (api.CommAbort) api.CommAbort(comm_);  // unblocks this rank's pending collectives; the peers see an RCCL error instead of waiting for us
        else api.CommDestroy(comm_);
        comm_ = nullptr;
        aborted_ = true;
    }
}

#undef G_HIP
#undef G_NCCL
// inside gather() a failure between the collectives must not leave the communicator half way through a sequence its peers are still in
#define G_HIP(call)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            poison();                                                                        \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)
#define G_NCCL(call)                                                                         \
    do {                                                                                     \
        int r_ = (call);                                                                     \
        if (r_ != 0) {                                                                       \
            err = std::string(#call) + ": " + RcclApi::get().GetErrorString(r_);             \
            poison();                                                                        \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)

int PcmGather::gather(const void* pcm, int64_t pcm_stride, const int64_t* lengths_host, hipStream_t producer, Result* out, std::string& err) {
    if (!out) {
        err = "vits_pcm_gather: null argument";
        return -1;
    }
    if (broken_) {
        err = "vits_pcm_gather: this gather object failed inside an exchange before (its communicator was aborted): destroy it and make a new one on every rank";
        return -1;
    }
    const RcclApi& api = RcclApi::get();
    const size_t n = (size_t)world_ * rows_;
    // this rank's own checks do NOT return: they travel as -1 rows, so that every rank reaches the same verdict (see verdict())
    const bool args_ok = pcm && lengths_host && pcm_stride >= 1;
    len_send_h_[0] = cap_;
    for (int b = 0; b < rows_; ++b) {
        int64_t l = args_ok ? lengths_host[b] : -1;
        if (l < 0 || l > pcm_stride || l > cap_) l = -1;
        len_send_h_[1 + b] = l;
    }
    if (producer && args_ok) {  // the exchange runs behind whatever wrote the PCM
        G_HIP(hipEventRecord(ev_, producer));
        G_HIP(hipStreamWaitEvent(side_, ev_, 0));
    }
    // (1) capacity + lengths: fixed size, so that every rank knows the verdict and the common row width of (2)
    const size_t tab = (size_t)rows_ + 1;
    if (comm_) {
        G_HIP(hipMemcpyAsync(len_send_d_, len_send_h_, sizeof(int64_t) * tab, hipMemcpyHostToDevice, side_));
        G_NCCL(api.AllGather(len_send_d_, len_all_d_, sizeof(int64_t) * tab, /*ncclInt8*/ 0, comm_, side_));
        G_HIP(hipMemcpyAsync(len_all_h_, len_all_d_, sizeof(int64_t) * tab * world_, hipMemcpyDeviceToHost, side_));
        G_HIP(hipStreamSynchronize(side_));
    } else
        std::copy(len_send_h_, len_send_h_ + tab, len_all_h_);
    int64_t smax = 1;
    if (verdict(len_all_h_, world_, rows_, &smax, err) != 0) return -1;  // (every rank returns here together: the communicator stays usable)
    for (int r = 0; r < world_; ++r) std::copy(len_all_h_ + (size_t)r * tab + 1, len_all_h_ + (size_t)(r + 1) * tab, len_out_h_.begin() + (size_t)r * rows_);
    // (2) the rows, padded to the longest utterance of any rank (an all-gather only copies: both element types travel as bytes). Only the
    // first min(smax, pcm_stride) elements of a local row are read — pcm_stride need only reach this rank's OWN longest row —; what lies
    // between a row's length and the common width is unspecified (include/vits.h says so).
    const size_t row_bytes = (size_t)smax * eb_;
    const size_t copy_bytes = (size_t)std::min(smax, pcm_stride) * eb_;
    G_HIP(hipMemcpy2DAsync(send_, row_bytes, pcm, (size_t)pcm_stride * eb_, copy_bytes, (size_t)rows_, hipMemcpyDeviceToDevice, side_));
    if (comm_) G_NCCL(api.AllGather(send_, out_, row_bytes * rows_, /*ncclInt8*/ 0, comm_, side_));
    else
        G_HIP(hipMemcpyAsync(out_, send_, row_bytes * rows_, hipMemcpyDeviceToDevice, side_));
    G_HIP(hipStreamSynchronize(side_));
    bytes_moved += (int64_t)(n * row_bytes);
    out->data = out_;
    out->stride = smax;
    out->lengths = len_out_h_.data();
    out->rows_total = (int32_t)n;
    return 0;
}

}  // namespace vits
