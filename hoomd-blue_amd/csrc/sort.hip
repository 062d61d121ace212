// sort.hip -- ascending sort of 64-bit keys on the device (rocPRIM's radix sort behind a C entry point).
//
// The exchange code builds its send lists as unordered keys (destination << 32 | particle index) appended by many
// waves at once; sorted, they are the send list grouped by destination with a reproducible order inside each group.
// The reference sorts the same way with thrust / hipCUB behind its CachedAllocator (hoomd/CommunicatorGPU.cu:374-520).
#include <cstring>

#include "common.cuh"

#include <rocprim/device/device_radix_sort.hpp>

extern "C" size_t hmd_sort_keys_workspace_bytes(unsigned int n)
    {
    size_t bytes = 0;
    const unsigned long long* in = nullptr;
    unsigned long long* out = nullptr;
    if (rocprim::radix_sort_keys(nullptr, bytes, in, out, (size_t)n, 0u, 64u, (hipStream_t)0) != hipSuccess)
        return 0;
    return bytes + 256;
    }

extern "C" int hmd_sort_keys(const unsigned long long* d_keys_in, unsigned long long* d_keys_out, unsigned int n, unsigned int key_bits,
                             void* d_workspace, size_t workspace_bytes, void* stream)
    {
    HMD_REQUIRE(n == 0 || (d_keys_in && d_keys_out && d_keys_in != d_keys_out), HMD_ERR_INVALID_ARGUMENT,
                "hmd_sort_keys: input and output are required and must differ");
    HMD_REQUIRE(key_bits >= 1 && key_bits <= 64, HMD_ERR_INVALID_ARGUMENT, "hmd_sort_keys: key_bits must be in [1, 64]");
    if (n == 0)
        return HMD_OK;
    size_t need = 0;
    if (rocprim::radix_sort_keys(nullptr, need, d_keys_in, d_keys_out, (size_t)n, 0u, key_bits, (hipStream_t)stream) != hipSuccess)
        return HMD_ERR_INVALID_ARGUMENT;
    HMD_REQUIRE(d_workspace && workspace_bytes >= need, HMD_ERR_INVALID_ARGUMENT, "hmd_sort_keys: workspace too small (hmd_sort_keys_workspace_bytes)");
    const hipError_t e = rocprim::radix_sort_keys(d_workspace, need, d_keys_in, d_keys_out, (size_t)n, 0u, key_bits, (hipStream_t)stream);
    return e == hipSuccess ? hmd_launch_status() : (int)e;
    }
