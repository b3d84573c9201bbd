// sddp_sort.hip -- see sddp_sort.hpp
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "sddp_sort.hpp"

namespace sddp {

hipError_t sort_pairs_desc_temp_bytes(int n, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs_desc(nullptr, *bytes, static_cast<const double*>(nullptr), static_cast<double*>(nullptr),
                                          static_cast<const int*>(nullptr), static_cast<int*>(nullptr), size_t(n), 0, 64, nullptr);
}

hipError_t sort_pairs_desc(void* tmp, size_t tmp_bytes, const double* key_in, double* key_out, const int* val_in, int* val_out, int n,
                           hipStream_t stream) {
    return rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, key_in, key_out, val_in, val_out, size_t(n), 0, 64, stream);
}

}  // namespace sddp
