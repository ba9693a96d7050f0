// pcm_gather.h — the path's ONE exchange for C / C++ / Swift hosts: the ragged all-gather of the PCM over RCCL (xGMI), behind the C ABI of
// include/vits.h (vits_pcm_gather_*). What vits.cpp_amd/multi_gpu.py does through torch.distributed, without Python: utterances are
// sharded across one process per GPU (the reference processes one utterance per call, /root/reference/src/vits.cpp:184,303, so shards
// never interact), and the only traffic is the per-utterance lengths followed by the padded rows.
// RCCL is loaded with dlopen on first use (single-GPU users need nothing); world == 1 never touches it unless asked to.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace vits {

struct RcclApi {
    // the five entry points used, with RCCL's own signatures (rccl.h:187,220,260,339,678); ncclUniqueId is 128 opaque bytes by value
    struct UniqueId {
        char internal[128];
    };
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, UniqueId id, int rank) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s) = nullptr;
    int (*CommAbort)(void* comm) = nullptr;  // optional (rccl.h ncclCommAbort): the failure path of PcmGather::gather
    std::string why;  // why it is not available
    bool ok() const { return AllGather != nullptr; }
    static const RcclApi& get();  // dlopen(VITS_RCCL_LIB or librccl.so.1 / librccl.so) once per process
};

class PcmGather {
  public:
    ~PcmGather();
    // joins the communicator on the CURRENT HIP device. rows: utterances per rank (the same on every rank: pad a short shard with
    // zero-length rows); capacity: elements per row every rank's buffer can hold; elem_bytes: 4 (fp32 PCM) or 2 (PCM16).
    bool init(const char* id, size_t id_bytes, int rank, int world, int rows, int64_t capacity, int elem_bytes, std::string& err);
    struct Result {
        const void* data = nullptr;  // device [world * rows][stride]
        int64_t stride = 0;
        const int64_t* lengths = nullptr;  // host [world * rows], rank blocks in rank order
        int32_t rows_total = 0;
    };
    // pcm: device [rows][pcm_stride] elements, row b valid up to lengths_host[b]; producer: the stream the PCM was written on (the
    // exchange is ordered behind it; nullptr = the caller has synchronised). Blocks until the gathered block is complete.
    // COLLECTIVE, also in failure: a rank whose own arguments are unusable still takes part in the first all-gather (its rows travel as -1)
    // and every rank returns -1 with the same verdict; elements of a gathered row beyond its length are unspecified.
    int gather(const void* pcm, int64_t pcm_stride, const int64_t* lengths_host, hipStream_t producer, Result* out, std::string& err);
    // the verdict on the table of the first all-gather ([row_capacity, lengths...] per rank): 0 + the common row width, or -1 + a message.
    // A pure function of the table (identical on every rank), exported as vits_pcm_gather_verdict for the CPU tests.
    static int verdict(const int64_t* table, int world, int rows, int64_t* smax_out, std::string& err);
    bool aborted() const { return aborted_; }
    int world() const { return world_; }
    bool uses_rccl() const { return comm_ != nullptr; }
    int64_t bytes_moved = 0;

  private:
    int rank_ = 0, world_ = 1, rows_ = 0, eb_ = 4;
    int64_t cap_ = 0;
    void* comm_ = nullptr;
    hipStream_t side_ = nullptr;
    hipEvent_t ev_ = nullptr;
    bool broken_ = false, aborted_ = false;
    void poison();  // a HIP / RCCL failure inside an exchange: abort the communicator (peers get an error, not a hang), refuse further calls
    std::vector<int64_t> len_out_h_;                         // [world * rows]: what Result::lengths points at
    int64_t *len_send_h_ = nullptr, *len_all_h_ = nullptr;  // pinned, [rows + 1] and [world][rows + 1]
    int64_t *len_send_d_ = nullptr, *len_all_d_ = nullptr;
    char *send_ = nullptr, *out_ = nullptr;
};

}  // namespace vits
