// Bucket grouping for the Pippenger MSM: radix sort of (window|bucket key, point index|sign) pairs.
// rocPRIM's onesweep radix sort is a library primitive (like a plain GEMM would be); everything around it is ours.
#include "common.cuh"
#include <rocprim/rocprim.hpp>

namespace zk {

int32_t sort_pairs_temp_bytes(zkpor_ctx* ctx, size_t n, int end_bit, size_t* bytes) {
    rocprim::double_buffer<u32> k(nullptr, nullptr);
    rocprim::double_buffer<u32> v(nullptr, nullptr);
    size_t tb = 0;
    ZK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, k, v, n, 0, (unsigned)end_bit, ctx->stream));
    *bytes = tb;
    return ZKPOR_OK;
}

int32_t sort_pairs(zkpor_ctx* ctx, void* temp, size_t temp_bytes, u32* k0, u32* k1, u32* v0, u32* v1, size_t n,
                   int end_bit, u32** k_out, u32** v_out) {
    rocprim::double_buffer<u32> k(k0, k1);
    rocprim::double_buffer<u32> v(v0, v1);
    ZK_HIP(ctx, rocprim::radix_sort_pairs(temp, temp_bytes, k, v, n, 0, (unsigned)end_bit, ctx->stream));
    *k_out = k.current();
    *v_out = v.current();
    return ZKPOR_OK;
}

}  // namespace zk
