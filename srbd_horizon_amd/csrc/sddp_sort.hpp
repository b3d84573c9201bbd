// sddp_sort.hpp -- descending key sort of the cold-queue order (sddp_options.queue_order = 2); implemented in sddp_sort.hip on
// rocPRIM's device radix sort (a separate translation unit: the solver kernels do not see the rocPRIM headers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace sddp {

// bytes of temporary device storage sort_pairs_desc needs for n pairs
hipError_t sort_pairs_desc_temp_bytes(int n, size_t* bytes);
// (key_out, val_out) = (key_in, val_in) sorted by key, largest first; asynchronous on `stream`
hipError_t sort_pairs_desc(void* tmp, size_t tmp_bytes, const double* key_in, double* key_out, const int* val_in, int* val_out, int n,
                           hipStream_t stream);

}  // namespace sddp
