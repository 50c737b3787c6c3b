// hostsim.h -- shared between the fake HIP runtime, the fake launchers and the driver (tests/hostsim/).
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <thread>

namespace hostsim {
struct Counters {
    std::atomic<long> mallocs{0}, frees{0}, graphs{0}, graph_launches{0};
    std::atomic<long> dc_launches{0}, finalize_launches{0}, tail_launches{0}, mfma_launches{0}, other_launches{0}, resident_starts{0}, resident_calls{0};
    std::atomic<long> violations{0}; // planner invariants broken (each one is printed)
};
extern Counters counters;
void attach_worker(hipStream_t s, std::thread &&t); // the emulated resident kernel of a launch on stream s
float resident_value(unsigned seq_independent_slot, int o); // what the emulated workgroup `slot` posts as its sum o
} // namespace hostsim
