// launch_ev.hpp — a completion event carried by a kernel's OWN dispatch packet.
//
// cream_block_bwd orders the weight-gradient stream behind kernels of the main chain.  An event recorded behind a
// kernel is a marker packet between it and the next kernel of the chain: 2.7 us of step time each (4.1 us with the
// default system-scope fence), 64 of them per training step (profiles/r04_step_gaps.md).  hipExtLaunchKernelGGL attaches
// the event to the kernel's dispatch packet instead: the packet's completion signal is the event, nothing is enqueued
// behind it.  The sequencing code arms `tl_stop_event` right before the call that launches the producing kernel; the
// launch helper of that kernel (CREAM_LAUNCH) consumes it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>

namespace cream {
extern thread_local hipEvent_t tl_stop_event;       // defined in block_seq.cpp
// in-step kernel timing (cream_block_prof_*): a start event for the same launch — the pair then brackets the kernel's own
// execution on its dispatch packet (what rocprofv3 reports), with no marker packets in the stream
extern thread_local hipEvent_t tl_start_event;
}

#define CREAM_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                        \
    do {                                                                                                             \
        hipEvent_t ev__ = cream::tl_stop_event, sv__ = cream::tl_start_event;                                        \
        if (ev__ || sv__) {                                                                                          \
            cream::tl_stop_event = nullptr;                                                                          \
            cream::tl_start_event = nullptr;                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, sv__, ev__, 0, __VA_ARGS__);                   \
        } else {                                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                     \
        }                                                                                                            \
    } while (0)

namespace cream {
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a KERNEL: remembered per (kernel address, device)
// — one process may drive several GPUs (block_seq.cpp: MAX_DEV), and kernels of the same signature (every gemm_nt8_kernel<EPI>
// is a void(*)(NtParams)) must not share a record: a static per template instantiation over the pointer TYPE did.
// A small open-addressed table of atomics: no lock on the launch path, a lost race only repeats the host-side call.
inline bool raise_dynamic_lds_addr(const void* kern, int bytes)
{
    constexpr int SLOTS = 256;                                 // (the library has ~40 kernels that ask for dynamic LDS)
    static std::atomic<const void*> key[SLOTS];
    static std::atomic<uint32_t> done[SLOTS];                  // bit d: raised on device d
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    int slot = -1;
    if (dev >= 0 && dev < 32) {
        uintptr_t h = reinterpret_cast<uintptr_t>(kern);
        h = (h >> 4) * 0x9E3779B97F4A7C15ull >> 40;
        for (int probe = 0; probe < SLOTS; ++probe) {
            const int s = (int)((h + probe) % SLOTS);
            const void* k = key[s].load(std::memory_order_acquire);
            if (k == nullptr) {
                const void* expect = nullptr;
                if (key[s].compare_exchange_strong(expect, kern, std::memory_order_acq_rel) || expect == kern) { slot = s; break; }
                continue;
            }
            if (k == kern) { slot = s; break; }
        }
        if (slot >= 0 && (done[slot].load(std::memory_order_acquire) >> dev & 1u)) return true;
    }
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (slot >= 0) done[slot].fetch_or(1u << dev, std::memory_order_release);
    return true;
}
template <typename K>
bool raise_dynamic_lds(K kern, int bytes) { return raise_dynamic_lds_addr(reinterpret_cast<const void*>(kern), bytes); }
}  // namespace cream
