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
    if (comm_) RcclApi::get().CommDestroy(comm_);
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
    G_HIP(hipHostMalloc((void**)&len_send_h_, sizeof(int64_t) * rows, hipHostMallocDefault));
    G_HIP(hipHostMalloc((void**)&len_all_h_, sizeof(int64_t) * n, hipHostMallocDefault));
    G_HIP(hipMalloc((void**)&len_send_d_, sizeof(int64_t) * rows));
    G_HIP(hipMalloc((void**)&len_all_d_, sizeof(int64_t) * n));
    G_HIP(hipMalloc((void**)&send_, (size_t)rows * capacity * eb_));
    G_HIP(hipMalloc((void**)&out_, n * (size_t)capacity * eb_));
    return true;
}

int PcmGather::gather(const void* pcm, int64_t pcm_stride, const int64_t* lengths_host, hipStream_t producer, Result* out, std::string& err) {
    const int fail = -1;
    if (!pcm || !lengths_host || !out || pcm_stride < 1) {
        err = "vits_pcm_gather: null argument";
        return -1;
    }
    const RcclApi& api = RcclApi::get();
    const size_t n = (size_t)world_ * rows_;
    for (int b = 0; b < rows_; ++b) {
        if (lengths_host[b] < 0 || lengths_host[b] > pcm_stride || lengths_host[b] > cap_) {
            err = "vits_pcm_gather: a row is longer than pcm_stride / the capacity agreed at init";
            return -1;
        }
        len_send_h_[b] = lengths_host[b];
    }
    if (producer) {  // the exchange runs behind whatever wrote the PCM
        G_HIP(hipEventRecord(ev_, producer));
        G_HIP(hipStreamWaitEvent(side_, ev_, 0));
    }
    // (1) the lengths: fixed size, so that every rank knows the common row width of (2)
    if (comm_) {
        G_HIP(hipMemcpyAsync(len_send_d_, len_send_h_, sizeof(int64_t) * rows_, hipMemcpyHostToDevice, side_));
        G_NCCL(api.AllGather(len_send_d_, len_all_d_, sizeof(int64_t) * rows_, /*ncclInt8*/ 0, comm_, side_));
        G_HIP(hipMemcpyAsync(len_all_h_, len_all_d_, sizeof(int64_t) * n, hipMemcpyDeviceToHost, side_));
        G_HIP(hipStreamSynchronize(side_));
    } else
        std::copy(len_send_h_, len_send_h_ + rows_, len_all_h_);
    int64_t smax = 1;
    for (size_t i = 0; i < n; ++i) smax = std::max(smax, len_all_h_[i]);
    if (smax > cap_ || smax > pcm_stride) {
        err = "vits_pcm_gather: another rank holds an utterance longer than this rank's buffer (all ranks must agree on row_capacity, and pcm_stride must reach it)";
        return -1;
    }
    // (2) the rows, padded to the longest utterance of any rank (an all-gather only copies: both element types travel as bytes)
    const size_t row_bytes = (size_t)smax * eb_;
    G_HIP(hipMemcpy2DAsync(send_, row_bytes, pcm, (size_t)pcm_stride * eb_, row_bytes, (size_t)rows_, hipMemcpyDeviceToDevice, side_));
    if (comm_) G_NCCL(api.AllGather(send_, out_, row_bytes * rows_, /*ncclInt8*/ 0, comm_, side_));
    else
        G_HIP(hipMemcpyAsync(out_, send_, row_bytes * rows_, hipMemcpyDeviceToDevice, side_));
    G_HIP(hipStreamSynchronize(side_));
    bytes_moved += (int64_t)(n * row_bytes);
    out->data = out_;
    out->stride = smax;
    out->lengths = len_all_h_;
    out->rows_total = (int32_t)n;
    return 0;
}

}  // namespace vits
