// Which store forms need a wait state before their data registers are overwritten on gfx950?  (DESIGN.md "A second one
// (round 2)")  tools/hwtest/wide_store.hip showed it for buffer_store_dwordx4 with a REGISTER scalar
// offset, the one form for which the compiler inserts no wait state.  This program runs the same experiment -- store good
// data, overwrite the first data registers in the very next instruction, count what reached memory -- for the other
// forms, each with NO wait state, with one (`s_nop 0`) and with two (`s_nop 1`, what the compiler inserts where it knows the hazard):
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/hwtest/wide_store_forms tools/hwtest/wide_store_forms.hip && tools/hwtest/wide_store_forms
//
//   buffer_store_dwordx2 / x3 / x4, scalar offset in a register        (8 bytes: no hazard documented; 12 / 16: the finding)
//   buffer_store_dwordx3 / x4, scalar offset 0                          (documented hazard; the compiler covers it: control)
//   global_store_dwordx2 / x3 / x4                                      (documented for > 8 bytes; the compiler covers it)
// Every lane writes only its own slot of `out`; all addresses are inside the array: nothing here can fault.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr int ITER = 32, THREADS = 256;

enum Form { BUF2_S = 0, BUF3_S, BUF4_S, BUF3_0, BUF4_0, GLB2, GLB3, GLB4, FORMS };
static const char *const form_name[FORMS] = {"buffer_store_dwordx2, register soffset", "buffer_store_dwordx3, register soffset",
                                             "buffer_store_dwordx4, register soffset", "buffer_store_dwordx3, soffset 0", "buffer_store_dwordx4, soffset 0",
                                             "global_store_dwordx2", "global_store_dwordx3", "global_store_dwordx4"};
static const int form_dwords[FORMS] = {2, 3, 4, 3, 4, 2, 3, 4};

#define FILL "v_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\ts_nop 4\n\t"
#define GAP ".if %5 == 1\n\ts_nop 0\n\t.endif\n\t.if %5 == 2\n\ts_nop 1\n\t.endif\n\t"
#define SPOIL "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\ts_waitcnt vmcnt(0)"

template <int F, int NOPS>
__global__ __launch_bounds__(THREADS) void store_kernel(uint32_t *out, unsigned bytes)
{
    const uint64_t base = (uint64_t)(uintptr_t)out;
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((uint32_t)base); rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffffu);
    rs.z = __builtin_amdgcn_readfirstlane(bytes); rs.w = __builtin_amdgcn_readfirstlane(0x00020000u);
    const unsigned lane_off = 16u * threadIdx.x;                    // 16-byte slots whatever the store's width
    for (int it = 0; it < ITER; ++it) {
        const unsigned row = __builtin_amdgcn_readfirstlane((unsigned)((blockIdx.x * ITER + it) * (THREADS * 16)));
        const unsigned good = 0x600D0000u | (unsigned)it, bad = 0xBAD0BAD0u;
        const unsigned vo = lane_off + row;
        uint32_t *gp = out + (size_t)vo / 4;
        if (F == BUF2_S)
            asm volatile(FILL "buffer_store_dwordx2 v[20:21], %1, %2, %3 offen\n\t" GAP SPOIL
                         :: "v"(good), "v"(lane_off), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        else if (F == BUF3_S)
            asm volatile(FILL "buffer_store_dwordx3 v[20:22], %1, %2, %3 offen\n\t" GAP SPOIL
                         :: "v"(good), "v"(lane_off), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        else if (F == BUF4_S)
            asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\t" GAP SPOIL
                         :: "v"(good), "v"(lane_off), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        else if (F == BUF3_0)
            asm volatile(FILL "buffer_store_dwordx3 v[20:22], %1, %2, 0 offen\n\t" GAP SPOIL
                         :: "v"(good), "v"(vo), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        else if (F == BUF4_0)
            asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\t" GAP SPOIL
                         :: "v"(good), "v"(vo), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        else if (F == GLB2)
            asm volatile(FILL "global_store_dwordx2 %1, v[20:21], off\n\t" GAP SPOIL
                         :: "v"(good), "v"(gp), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        else if (F == GLB3)
            asm volatile(FILL "global_store_dwordx3 %1, v[20:22], off\n\t" GAP SPOIL
                         :: "v"(good), "v"(gp), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        else
            asm volatile(FILL "global_store_dwordx4 %1, v[20:23], off\n\t" GAP SPOIL
                         :: "v"(good), "v"(gp), "s"(rs), "s"(row), "v"(bad), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
    }
}

template <int F, int NOPS>
static long run(uint32_t *d_out, size_t dwords, int blocks, std::vector<uint32_t> &h)
{
    CHECK(hipMemset(d_out, 0, dwords * 4));
    hipLaunchKernelGGL((store_kernel<F, NOPS>), dim3(blocks), dim3(THREADS), 0, 0, d_out, (unsigned)(dwords * 4));
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d_out, dwords * 4, hipMemcpyDeviceToHost));
    long wrong = 0;
    for (size_t i = 0; i < dwords; ++i) {
        if ((int)(i & 3) >= form_dwords[F]) continue;                 // the slot's tail a narrower store leaves alone
        const unsigned it = (unsigned)((i / (THREADS * 4)) % ITER);
        wrong += h[i] != (0x600D0000u | it);
    }
    return wrong;
}

template <int F>
static void form(uint32_t *d_out, size_t dwords, int blocks, std::vector<uint32_t> &h)
{
    const long n0 = run<F, 0>(d_out, dwords, blocks, h), n1 = run<F, 1>(d_out, dwords, blocks, h), n2 = run<F, 2>(d_out, dwords, blocks, h);
    printf("  %-42s no wait state: %9ld wrong   s_nop 0: %9ld   s_nop 1: %9ld   -> %s\n", form_name[F], n0, n1, n2,
           n0 ? (n1 ? (n2 ? "needs MORE than two wait states" : "needs two wait states") : "needs a wait state") : "no hazard seen");
}

int main()
{
    const int sizes[] = {256, 4096};                    // one and sixteen workgroups per CU
    for (int blocks : sizes) {
        const size_t dwords = (size_t)blocks * ITER * THREADS * 4;
        uint32_t *d_out;
        CHECK(hipMalloc(&d_out, dwords * 4));
        std::vector<uint32_t> h(dwords);
        printf("%d workgroups: a store followed at once by a write to its first two data registers\n", blocks);
        form<BUF2_S>(d_out, dwords, blocks, h); form<BUF3_S>(d_out, dwords, blocks, h); form<BUF4_S>(d_out, dwords, blocks, h);
        form<BUF3_0>(d_out, dwords, blocks, h); form<BUF4_0>(d_out, dwords, blocks, h);
        form<GLB2>(d_out, dwords, blocks, h); form<GLB3>(d_out, dwords, blocks, h); form<GLB4>(d_out, dwords, blocks, h);
        CHECK(hipFree(d_out));
    }
    return 0;
}
