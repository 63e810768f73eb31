// Bucket grouping for the Pippenger MSM: radix sort of (window|bucket key, point index|sign) pairs.
// rocPRIM's onesweep radix sort is a library primitive (like a plain GEMM would be); everything around it is ours.
//
// The sort runs on the auxiliary stream UNDER the NTT passes and the bucket accumulations of the main stream.  rocPRIM's default
// onesweep kernel for 4-byte pairs uses 1024-thread workgroups: next to a long-running kernel whose 256-thread workgroups trickle
// out one at a time (and are replaced at once by the next workgroup of the same kernel) a 16-wave workgroup almost never finds a CU
// with 16 free wave slots, so a 3 ms pass took 70-90 ms and the main stream ended up waiting for it (profiles/r03_timeline.txt).
// "sort_block" selects a configuration with 256- or 512-thread workgroups that compete for freed slots on equal terms.
#include "common.cuh"
#include <rocprim/rocprim.hpp>

namespace zk {

namespace {
using cfg256 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                          rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<256, 16>, 8,
                                                                              rocprim::block_radix_rank_algorithm::match>>;
using cfg512 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                          rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<512, 12>, 8,
                                                                              rocprim::block_radix_rank_algorithm::match>>;

template <class Cfg>
hipError_t run(void* temp, size_t& tb, rocprim::double_buffer<u32>& k, rocprim::double_buffer<u32>& v, size_t n, int end_bit, hipStream_t s) {
    return rocprim::radix_sort_pairs<Cfg>(temp, tb, k, v, n, 0, (unsigned)end_bit, s);
}
hipError_t dispatch(int block, void* temp, size_t& tb, rocprim::double_buffer<u32>& k, rocprim::double_buffer<u32>& v, size_t n, int end_bit, hipStream_t s) {
    if (block == 256) return run<cfg256>(temp, tb, k, v, n, end_bit, s);
    if (block == 512) return run<cfg512>(temp, tb, k, v, n, end_bit, s);
    return run<rocprim::default_config>(temp, tb, k, v, n, end_bit, s);
}
}  // namespace

int32_t sort_pairs_temp_bytes(zkpor_ctx* ctx, size_t n, int end_bit, size_t* bytes) {
    size_t best = 0;
    for (int block : {0, 256, 512}) {   // the workspace is sized once: take the largest of the selectable configurations
        rocprim::double_buffer<u32> k(nullptr, nullptr);
        rocprim::double_buffer<u32> v(nullptr, nullptr);
        size_t tb = 0;
        ZK_HIP(ctx, dispatch(block, nullptr, tb, k, v, n, end_bit, ctx->stream));
        if (tb > best) best = tb;
    }
    *bytes = best;
    return ZKPOR_OK;
}

int32_t sort_pairs(zkpor_ctx* ctx, void* temp, size_t temp_bytes, u32* k0, u32* k1, u32* v0, u32* v1, size_t n,
                   int end_bit, u32** k_out, u32** v_out) {
    rocprim::double_buffer<u32> k(k0, k1);
    rocprim::double_buffer<u32> v(v0, v1);
    ZK_HIP(ctx, dispatch(ctx->sort_block, temp, temp_bytes, k, v, n, end_bit, ctx->stream));
    *k_out = k.current();
    *v_out = v.current();
    return ZKPOR_OK;
}

}  // namespace zk
