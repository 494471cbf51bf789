// Stand-alone reproduction of the hazard behind DESIGN.md "A second one (round 2)": a 16-byte buffer store whose scalar
// offset is a REGISTER, followed by a vector ALU write to its first data registers.  The compiler keeps no distance
// between the two for this form of the store; the question here is whether gfx950 needs one, and how many wait states.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/hwtest/wide_store tools/hwtest/wide_store.hip && tools/hwtest/wide_store
//
// Every lane stores 16 bytes of 0x600D0000 | iteration to its own slot of `out` (all addresses inside the array:
// nothing here can fault) and overwrites two of the four data registers with 0xBAD0BAD0 right behind the store, with
// NOPS wait states between the two (-1: the same store with the whole offset in the vector register and the scalar
// offset 0, the form for which the compiler does insert the wait state; here too the wait state is left out on
// purpose, as the control).  The host counts the dwords that hold anything but the good pattern.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int ITER = 32, THREADS = 256;

template <int NOPS>
__global__ __launch_bounds__(THREADS) void store_kernel(uint32_t *out, unsigned bytes)
{
    // the array as a raw buffer: base, no stride, `bytes` records, 32-bit data format
    const uint64_t base = (uint64_t)(uintptr_t)out;
    u32x4 rs;
    rs.x = (uint32_t)base; rs.y = (uint32_t)(base >> 32) & 0xffffu; rs.z = bytes; rs.w = 0x00020000u;
    rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y);
    rs.z = __builtin_amdgcn_readfirstlane(rs.z); rs.w = __builtin_amdgcn_readfirstlane(rs.w);
    const unsigned lane_off = 16u * threadIdx.x;                                   // within the workgroup's row
    for (int it = 0; it < ITER; ++it) {
        // slot of (workgroup, iteration): uniform -> scalar offset
        const unsigned row = __builtin_amdgcn_readfirstlane((unsigned)((blockIdx.x * ITER + it) * (THREADS * 16)));
        const unsigned good = 0x600D0000u | (unsigned)it, bad = 0xBAD0BAD0u;
        if (NOPS < 0) {
            const unsigned vo = lane_off + row;
            asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\t"
                         "v_mov_b32 v20, %3\n\tv_mov_b32 v21, %3\n\t"
                         "s_waitcnt vmcnt(0)"
                         :: "v"(good), "v"(vo), "s"(rs), "v"(bad) : "v20", "v21", "v22", "v23", "memory");
        } else {
            asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\t"
                         ".if %5 == 1\n\ts_nop 0\n\t.endif\n\t.if %5 == 2\n\ts_nop 1\n\t.endif\n\t.if %5 == 3\n\ts_nop 2\n\t.endif\n\t"
                         "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\t"
                         "s_waitcnt vmcnt(0)"
                         :: "v"(good), "v"(lane_off), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        }
    }
}

template <int NOPS>
static long run(uint32_t *d_out, size_t dwords, int blocks, std::vector<uint32_t> &h)
{
    CHECK(hipMemset(d_out, 0, dwords * 4));
    hipLaunchKernelGGL(store_kernel<NOPS>, dim3(blocks), dim3(THREADS), 0, 0, d_out, (unsigned)(dwords * 4));
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d_out, dwords * 4, hipMemcpyDeviceToHost));
    long wrong = 0;
    for (size_t i = 0; i < dwords; ++i) {
        const unsigned it = (unsigned)((i / (THREADS * 4)) % ITER);
        wrong += h[i] != (0x600D0000u | it);
    }
    return wrong;
}

int main()
{
    const int sizes[] = {64, 256, 1024, 4096};          // workgroups: a quarter of the CUs ... 16 per CU
    printf("wait states between a 16-byte buffer store and a write to its data registers: wrong dwords of the total\n");
    for (int blocks : sizes) {
        const size_t dwords = (size_t)blocks * ITER * THREADS * 4;      // at most 1 GiB / 4
        uint32_t *d_out;
        CHECK(hipMalloc(&d_out, dwords * 4));
        std::vector<uint32_t> h(dwords);
        const long c = run<-1>(d_out, dwords, blocks, h), n0 = run<0>(d_out, dwords, blocks, h), n1 = run<1>(d_out, dwords, blocks, h),
                   n2 = run<2>(d_out, dwords, blocks, h), n3 = run<3>(d_out, dwords, blocks, h);
        printf("%5d workgroups (%zu dwords): scalar offset 0, no wait state %ld | register scalar offset: none %ld, s_nop 0 %ld, s_nop 1 %ld, s_nop 2 %ld\n",
               blocks, dwords, c, n0, n1, n2, n3);
        CHECK(hipFree(d_out));
    }
    return 0;
}
