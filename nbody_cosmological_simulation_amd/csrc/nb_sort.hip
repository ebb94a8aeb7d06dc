// nb_sort.hip -- stable key / key-value radix sorts for the diagnostics (nb_metrics.hip).
//
// The reference ranks the stars with torch.sort / torch.argsort (metrics.py:93, :128).  The sort itself is a plain
// library operation, so it comes from rocPRIM's device radix sort (header-only, part of ROCm); everything around it
// (keys, scan, decisions) is in nb_metrics.hip.  Radix sorts are stable: equal keys keep their index order, which is
// the tie rule the diagnostics document.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "nb_internal.h"

namespace {
template <typename K>
hipError_t sort_keys(void *tmp, size_t &bytes, const void *kin, void *kout, int n, hipStream_t st)
{
    return rocprim::radix_sort_keys(tmp, bytes, (const K *)kin, (K *)kout, (size_t)n, 0u, (unsigned)(8 * sizeof(K)), st);
}
template <typename K>
hipError_t sort_pairs(void *tmp, size_t &bytes, const void *kin, void *kout, const int *vin, int *vout, int n, hipStream_t st)
{
    return rocprim::radix_sort_pairs(tmp, bytes, (const K *)kin, (K *)kout, vin, vout, (size_t)n, 0u,
                                     (unsigned)(8 * sizeof(K)), st);
}
}  // namespace

size_t nb_sort_temp_bytes(int n, int key64)
{
    size_t a = 0, b = 0;
    const hipError_t ea = key64 ? sort_keys<unsigned long long>(nullptr, a, nullptr, nullptr, n, nullptr)
                                : sort_keys<unsigned>(nullptr, a, nullptr, nullptr, n, nullptr);
    const hipError_t eb = key64 ? sort_pairs<unsigned long long>(nullptr, b, nullptr, nullptr, nullptr, nullptr, n, nullptr)
                                : sort_pairs<unsigned>(nullptr, b, nullptr, nullptr, nullptr, nullptr, n, nullptr);
    if (ea != hipSuccess || eb != hipSuccess) return 0;        // the launch reports the error
    return (a > b ? a : b) + 256;
}

hipError_t nb_sort_keys(void *tmp, size_t tmp_bytes, const void *kin, void *kout, int n, int key64, hipStream_t st)
{
    if (!tmp || tmp_bytes == 0) return hipErrorInvalidValue;
    return key64 ? sort_keys<unsigned long long>(tmp, tmp_bytes, kin, kout, n, st)
                 : sort_keys<unsigned>(tmp, tmp_bytes, kin, kout, n, st);
}

hipError_t nb_sort_pairs(void *tmp, size_t tmp_bytes, const void *kin, void *kout, const int *vin, int *vout, int n,
                         int key64, hipStream_t st)
{
    if (!tmp || tmp_bytes == 0) return hipErrorInvalidValue;
    return key64 ? sort_pairs<unsigned long long>(tmp, tmp_bytes, kin, kout, vin, vout, n, st)
                 : sort_pairs<unsigned>(tmp, tmp_bytes, kin, kout, vin, vout, n, st);
}
