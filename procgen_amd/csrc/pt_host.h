// Host-side helpers shared by the translation units of libgpu_compute.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

namespace pt {

// thread-local last-error slot behind pt_last_error()/pt_last_error_string()
int set_error(int code, const std::string &msg);
int clear_error();

// SHA-256 of a byte string (FIPS 180-4); the seed-to-key rule lives in pt_seed_key().
void sha256(const uint8_t *data, size_t len, uint8_t out[32]);

// HIP runtime resolved and at least one device present
int device_count();

}  // namespace pt

struct pt_graph_engine;
// One iteration of the drop-in entry points on an engine that keeps the chains resident between calls (not
// part of the C ABI): see pt_graph.hip.
int pt_graph_legacy_iteration(pt_graph_engine *e, void *positions, void *costs, int real_bytes, long long call_index,
                              bool positions_are_new);

#define PT_HIP_TRY(expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return pt::set_error(_e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP,  \
                                 std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)
