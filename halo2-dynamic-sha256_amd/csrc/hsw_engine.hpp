// hsw_engine.hpp -- the engine object behind `hsw_engine *` and two helpers shared by the translation
// units that implement the C ABI (hsw_api.cpp: blocks, packing, host delivery; hsw_api_region.cpp: digest
// frames, constraint structure, on-device verification).  Internal: not part of the boundary.
#ifndef HSW_ENGINE_HPP
#define HSW_ENGINE_HPP

#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/hsw.h"
#include "hsw_frame.hpp"
#include "hsw_verify.h"

struct hsw_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hsw_shape shape{};
    int limbs = 2;
    int parts = 0;             // waves per block; 0 = choose from the batch size
    int helpers = 0;           // small-batch kernel, Montgomery cells: waves per role, 0 = chosen from the batch size
    int split = -1;            // -1 = small-batch kernel for <= 32 blocks, 0 = never, 1 = one phase per wave (32 waves per
                               // block) in hsw_expand_kernel, 2 = small-batch kernel always
    int tile = 0;              // tile width in cells: 0 = choose, 32, 64 or 128
    int mont_emit = 1;         // Montgomery cells of the streaming kernel: converted at emit time (Em::M32) -- 0 = never,
                               // 1 = in default mode (where it wins), 2 = also in internals mode; else at write-out
    size_t chunk_blocks = (size_t)1 << 20;   // blocks per launch (longer batches are consecutive launches)
    uint32_t mode = HSW_MODE_DEFAULT;
    bool timing = false;
    bool timed = false;        // ev0/ev1 bracket a launch
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    // host-delivery pipeline (hsw_witness_blocks_host): two device staging slots,
    // kernel on `stream`, D2H on `copy_stream`
    hipStream_t copy_stream = nullptr;
    struct Slot {
        void *gate = nullptr, *cd = nullptr, *cs = nullptr;
        hipEvent_t kernel_done = nullptr, copy_done = nullptr;
    } slot[2];
    size_t slot_blocks = 0, slot_rows = 0;
    // digest frames (hsw_witness_frames): descriptors staged per call, and k^-1 for
    // k = 0..inv_n-1 in canonical ([0]) and Montgomery ([1]) form
    // descriptor staging: a ring of pinned, device-mapped host buffers the kernel reads directly
    // (no H2D copy, no stream sync unless four frame launches are already in flight)
    struct FrameSlot {
        hsw::FrameDesc *h = nullptr;
        hsw::FrameDesc *d = nullptr;     // device address of h (pinned, device-mapped)
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool inflight = false;
    } frame_slot[4];
    unsigned frame_next = 0;
    uint64_t *d_inv_tbl[2] = {nullptr, nullptr};
    size_t inv_n = 0;
    // on-device verification (hsw_verify_blocks): the block structure, uploaded on first use
    void *d_structure = nullptr;
    hsw::VerifyParams verify_tpl{};      // structure pointers / counts filled in
    uint64_t verify_checks_per_block = 0;
    int verify_slices = 0;               // workgroups per block in hsw_verify_kernel; 0 = default
    hsw::VerifyReport *d_report = nullptr;
    hsw_launch_info last_launch{};       // hsw_last_launch
    void *d_mont_tab = nullptr;          // Em::M32 kernels: Montgomery form of every byte, its spread, and the byte << 8 (24 KiB)
};

inline int set_err(hsw_engine *e, int status, const char *what, hipError_t he = hipSuccess) {
    if (e) {
        char buf[256];
        if (he != hipSuccess)
            std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(he));
        else
            std::snprintf(buf, sizeof buf, "%s", what);
        e->err = buf;
    }
    return status;
}

namespace hsw { struct SmallFrames; }
bool hsw_small_eligible(const hsw_engine *e, size_t n_blocks);
int hsw_witness_blocks_impl(hsw_engine *e, const hsw_witness_args *args, const hsw::SmallFrames *frames,
                            uint32_t *host_next_states);
int hsw_witness_digests_impl(hsw_engine *e, const hsw_digests_args *args, uint32_t *dev_next_states);

// Makes the engine's device current for the scope of one call (a no-op when it already is: the usual case,
// and these scopes nest three deep on the latency-critical path of a small digest).
struct DeviceScope {
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) { ok = true; prev = -1; }
        else ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

#endif
