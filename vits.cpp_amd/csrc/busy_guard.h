// busy_guard.h — "one call at a time per model handle", enforced. The reference has the same contract and does not check it
// (vits_model::process writes member tensors and the model's prefix stack: /root/reference/src/include/vits.h:22-30,
// src/vits_model_data.cpp:136-139); here a thread that enters a handle while another call on it is in progress — or a callback that
// re-enters its own handle — gets an error instead of a race on the arenas. Header-only and HIP-free so that it has a CPU unit test
// (tests/test_abi.py::test_busy_guard_admits_one_caller).
#pragma once
#include <atomic>

namespace vits {

class BusyGuard {
  public:
    // tries to take the flag; entered() tells whether this guard owns it (and will release it)
    explicit BusyGuard(std::atomic<bool>* flag) : flag_(flag) {
        if (!flag_) return;
        bool expected = false;
        entered_ = flag_->compare_exchange_strong(expected, true, std::memory_order_acquire);
    }
    BusyGuard(const BusyGuard&) = delete;
    BusyGuard& operator=(const BusyGuard&) = delete;
    ~BusyGuard() {
        if (entered_) flag_->store(false, std::memory_order_release);
    }
    bool entered() const { return entered_; }

  private:
    std::atomic<bool>* flag_;
    bool entered_ = false;
};

}  // namespace vits
