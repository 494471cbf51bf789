// Which forms of a 64-bit shift does the gfx950 "last register of the allocation" hazard reach?  (DESIGN.md "A toolchain
// hazard found by the parity tests")  tools/hwtest/seg_loop.hip showed it for v_lshlrev_b64 with the
// shift AMOUNT in the last VGPR of the wave's allocation; this program places operands there on purpose, one form per kernel:
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/hwtest/shift_forms tools/hwtest/shift_forms.hip && tools/hwtest/shift_forms
//
// Every kernel keeps exactly 16 VGPRs allocated (v15 is named in the asm and nothing above it is used: the build prints
// next_free_vgpr, which must be 16 for the placement to mean anything -- tools/hwtest/check_shift_forms.py asserts it from
// the ISA), loads a 64-bit value and a shift amount per lane, applies ONE shift instruction of the form under test with the
// operand in question in v15 / v[14:15] -- written by the instruction right before the shift (GAP 0) or two wait states
// earlier (GAP 1) -- and stores the result.  The host compares with the shift done in C.  All addresses
// are inside the arrays; nothing here can fault.  Many waves must be resident for the hazard to show (it reads a register of
// the neighbouring allocation), hence the large grid and the repetitions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

enum Form {
    F_LSHL_AMT_LAST = 0,    // v_lshlrev_b64 d, v15, data            amount in the LAST register   (the known failure)
    F_LSHR_AMT_LAST,        // v_lshrrev_b64 d, v15, data
    F_ASHR_AMT_LAST,        // v_ashrrev_i64 d, v15, data
    F_LSHL_AMT_NOT_LAST,    // v_lshlrev_b64 d, v14, data            amount in the last but one    (control: must pass)
    F_LSHL_DATA_LAST,       // v_lshlrev_b64 d, amt, v[14:15]        DATA pair in the last two registers
    F_LSHL_SGPR_AMT,        // v_lshlrev_b64 d, s20, v[14:15]        amount in an SGPR (uniform), data pair last
    F_LSHL_CONST_AMT,       // v_lshlrev_b64 d, 5, v[14:15]          constant amount, data pair last
    F_LSHL_ADD_SRC_LAST,    // v_lshl_add_u64 d, v[14:15], 3, add    shifted source in the last pair
    F_LSHL_ADD_ADD_LAST,    // v_lshl_add_u64 d, src, 3, v[14:15]    addend in the last pair
    F_LSHL_DST_LAST,        // v_lshlrev_b64 v[14:15], amt, data     DESTINATION pair last
    F_LSHL_CONST_DATA,      // v_lshlrev_b64 d, v15, -1              amount last, the shifted value an INLINE CONSTANT (seg_loop's instruction)
    F_LSHR_CONST_DATA,      // v_lshrrev_b64 d, v15, -1
    F_LSHL_CONST_DATA_NL,   // v_lshlrev_b64 d, v14, -1              the same with the amount in the last but one (control)
    F_ASHR_CONST_DATA,      // v_ashrrev_i64 d, v15, -2
    F_LSHL_SGPR_DATA,       // v_lshlrev_b64 d, v15, s[20:21]        the shifted value in an SGPR pair
    F_COUNT
};
static const char *const form_name[F_COUNT] = {
    "v_lshlrev_b64  amount = last VGPR", "v_lshrrev_b64  amount = last VGPR", "v_ashrrev_i64  amount = last VGPR",
    "v_lshlrev_b64  amount = last-but-one VGPR (control)", "v_lshlrev_b64  data pair = last two VGPRs",
    "v_lshlrev_b64  SGPR amount, data pair last", "v_lshlrev_b64  constant amount, data pair last",
    "v_lshl_add_u64 shifted source = last pair", "v_lshl_add_u64 addend = last pair", "v_lshlrev_b64  destination = last pair",
    "v_lshlrev_b64  amount = last VGPR, data = constant -1", "v_lshrrev_b64  amount = last VGPR, data = constant -1",
    "v_lshlrev_b64  amount = last-but-one, data = constant -1", "v_ashrrev_i64  amount = last VGPR, data = constant -2",
    "v_lshlrev_b64  amount = last VGPR, data = SGPR pair"};

// LD 1: a load of the lane's own amount word into v12 is issued right before and left in flight across the shift (the
// failing loops of seg_loop.hip have dependent loads in flight; its variants without loads do not fail)
#define LOADP ".if %c[ld] == 1\n\tglobal_load_dword v12, %[lp], off\n\t.endif\n\t"
#define LOADP14 ".if %c[ld] == 1\n\tglobal_load_dword v14, %[lp], off\n\t.endif\n\t"      // into the register next to the amount
#define WAITP "\n\ts_waitcnt vmcnt(0)"
template <int F, int GAP, int LD>
__global__ __launch_bounds__(256) void shift_kernel(const uint64_t *in, const int *amt, uint64_t *out, size_t n, int uniform_amt)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint64_t x = in[g];
    const int a = amt[g];
    const uint64_t y = in[n - 1 - g];                  // second 64-bit operand of the add forms
    uint64_t r = 0;
    if (F == F_LSHL_AMT_LAST)
        asm volatile(LOADP "v_mov_b32 v15, %2\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshlrev_b64 %0, v15, %1" WAITP : "=v"(r) : "v"(x), "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v15", "memory");
    else if (F == F_LSHR_AMT_LAST)
        asm volatile(LOADP "v_mov_b32 v15, %2\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshrrev_b64 %0, v15, %1" WAITP : "=v"(r) : "v"(x), "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v15", "memory");
    else if (F == F_ASHR_AMT_LAST)
        asm volatile(LOADP "v_mov_b32 v15, %2\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_ashrrev_i64 %0, v15, %1" WAITP : "=v"(r) : "v"(x), "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v15", "memory");
    else if (F == F_LSHL_AMT_NOT_LAST)
        asm volatile(LOADP "v_mov_b32 v14, %2\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshlrev_b64 %0, v14, %1" WAITP : "=v"(r) : "v"(x), "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v14", "v15", "memory");
    else if (F == F_LSHL_DATA_LAST)
        asm volatile(LOADP "v_mov_b32 v14, %1\n\tv_mov_b32 v15, %2\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshlrev_b64 %0, %3, v[14:15]" WAITP
                     : "=v"(r) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v14", "v15", "memory");
    else if (F == F_LSHL_SGPR_AMT)
        asm volatile(LOADP "v_mov_b32 v14, %1\n\tv_mov_b32 v15, %2\n\ts_mov_b32 s20, %3\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshlrev_b64 %0, s20, v[14:15]" WAITP
                     : "=v"(r) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "s"(uniform_amt), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v14", "v15", "s20", "memory");
    else if (F == F_LSHL_CONST_AMT)
        asm volatile(LOADP "v_mov_b32 v14, %1\n\tv_mov_b32 v15, %2\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshlrev_b64 %0, 5, v[14:15]" WAITP
                     : "=v"(r) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v14", "v15", "memory");
    else if (F == F_LSHL_ADD_SRC_LAST)
        asm volatile(LOADP "v_mov_b32 v14, %1\n\tv_mov_b32 v15, %2\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshl_add_u64 %0, v[14:15], 3, %3" WAITP
                     : "=v"(r) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "v"(y), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v14", "v15", "memory");
    else if (F == F_LSHL_ADD_ADD_LAST)
        asm volatile(LOADP "v_mov_b32 v14, %1\n\tv_mov_b32 v15, %2\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshl_add_u64 %0, %3, 3, v[14:15]" WAITP
                     : "=v"(r) : "v"((uint32_t)y), "v"((uint32_t)(y >> 32)), "v"(x), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v14", "v15", "memory");
    else if (F == F_LSHL_CONST_DATA)
        asm volatile(LOADP14 "v_mov_b32 v15, %1\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshlrev_b64 %0, v15, -1" WAITP : "=v"(r) : "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v14", "v15", "memory");
    else if (F == F_LSHR_CONST_DATA)
        asm volatile(LOADP14 "v_mov_b32 v15, %1\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshrrev_b64 %0, v15, -1" WAITP : "=v"(r) : "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v14", "v15", "memory");
    else if (F == F_LSHL_CONST_DATA_NL)
        asm volatile(LOADP "v_mov_b32 v14, %1\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshlrev_b64 %0, v14, -1" WAITP : "=v"(r) : "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v14", "v15", "memory");
    else if (F == F_ASHR_CONST_DATA)
        asm volatile(LOADP14 "v_mov_b32 v15, %1\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_ashrrev_i64 %0, v15, -2" WAITP : "=v"(r) : "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v14", "v15", "memory");
    else if (F == F_LSHL_SGPR_DATA)
        asm volatile(LOADP14 "v_mov_b32 v15, %1\n\ts_mov_b32 s20, 0x89abcdef\n\ts_mov_b32 s21, 0x01234567\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_lshlrev_b64 %0, v15, s[20:21]" WAITP
                     : "=v"(r) : "v"(a), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v14", "v15", "s20", "s21", "memory");
    else if (F == F_LSHL_DST_LAST) {
        uint32_t lo, hi;
        asm volatile(LOADP "v_lshlrev_b64 v[14:15], %2, %3\n\t.if %c[gap] == 1\n\ts_nop 1\n\t.endif\n\tv_mov_b32 %0, v14\n\tv_mov_b32 %1, v15" WAITP
                     : "=v"(lo), "=v"(hi) : "v"(a), "v"(x), [gap] "n"(GAP), [ld] "n"(LD), [lp] "v"(amt + g) : "v12", "v14", "v15", "memory");
        r = ((uint64_t)hi << 32) | lo;
    }
    out[g] = r;
}

static uint64_t expect(int f, uint64_t x, uint64_t y, int a, int ua)
{
    switch (f) {
    case F_LSHL_AMT_LAST: case F_LSHL_AMT_NOT_LAST: case F_LSHL_DATA_LAST: case F_LSHL_DST_LAST: return x << (a & 63);
    case F_LSHR_AMT_LAST: return x >> (a & 63);
    case F_ASHR_AMT_LAST: return (uint64_t)((int64_t)x >> (a & 63));
    case F_LSHL_SGPR_AMT: return x << (ua & 63);
    case F_LSHL_CONST_AMT: return x << 5;
    case F_LSHL_ADD_SRC_LAST: return (x << 3) + y;
    case F_LSHL_ADD_ADD_LAST: return (x << 3) + y;
    case F_LSHL_CONST_DATA: case F_LSHL_CONST_DATA_NL: return ~0ull << (a & 63);
    case F_LSHR_CONST_DATA: return ~0ull >> (a & 63);
    case F_ASHR_CONST_DATA: return (uint64_t)((int64_t)-2 >> (a & 63));
    case F_LSHL_SGPR_DATA: return 0x0123456789abcdefull << (a & 63);
    }
    return 0;
}

template <int F>
static void launch(int gap, dim3 grid, const uint64_t *in, const int *amt, uint64_t *out, size_t n, int ua)
{
    if (gap == 1) hipLaunchKernelGGL((shift_kernel<F, 1, 0>), grid, dim3(256), 0, 0, in, amt, out, n, ua);
    else if (gap == 0) hipLaunchKernelGGL((shift_kernel<F, 0, 0>), grid, dim3(256), 0, 0, in, amt, out, n, ua);
    else hipLaunchKernelGGL((shift_kernel<F, 0, 1>), grid, dim3(256), 0, 0, in, amt, out, n, ua);
}

int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atol(argv[1]) : 16) << 20;      // lanes: 16 M by default (4096 workgroups per CU-round)
    const int reps = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<uint64_t> h(n), o(n);
    std::vector<int> a(n);
    srand(3);
    for (size_t i = 0; i < n; ++i) {
        h[i] = ((uint64_t)rand() << 40) ^ ((uint64_t)rand() << 20) ^ (uint64_t)rand() ^ 0x8000000000000001ull;
        a[i] = rand() % 64;
    }
    uint64_t *d_in, *d_out; int *d_amt;
    CK(hipMalloc(&d_in, n * 8)); CK(hipMalloc(&d_out, n * 8)); CK(hipMalloc(&d_amt, n * 4));
    CK(hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_amt, a.data(), n * 4, hipMemcpyHostToDevice));
    const dim3 grid((unsigned)((n + 255) / 256));
    const int ua = 13;
    printf("64-bit shift forms with an operand in the last registers of a 16-VGPR allocation: wrong results of %zu lanes x %d launches\n", n, reps);
    for (int gap = 0; gap < 3; ++gap)
    for (int f = 0; f < F_COUNT; ++f) {
        long wrong = 0; int shown = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(d_out, 0xee, n * 8));
            switch (f) {
            case 0: launch<0>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 1: launch<1>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 2: launch<2>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 3: launch<3>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 4: launch<4>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 5: launch<5>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 6: launch<6>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 7: launch<7>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 8: launch<8>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 9: launch<9>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 10: launch<10>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 11: launch<11>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 12: launch<12>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 13: launch<13>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            case 14: launch<14>(gap, grid, d_in, d_amt, d_out, n, ua); break;
            }
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(o.data(), d_out, n * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; ++i) {
                // the add forms: F7 shifts x (this lane's value) and adds y; F8 shifts x held in ordinary registers and adds
                // y held in v[14:15] -- the same arithmetic, different register placement
                const uint64_t e = expect(f, h[i], h[n - 1 - i], a[i], ua);
                if (o[i] != e) {
                    if (shown < 2) { printf("    lane %zu: got %016llx want %016llx (amount %d)\n", i, (unsigned long long)o[i], (unsigned long long)e, a[i]); ++shown; }
                    ++wrong;
                }
            }
        }
        printf("  %-52s %-34s %s (%ld wrong)\n", form_name[f], gap == 1 ? "two wait states before the shift" : (gap == 0 ? "operand written just before" : "a load in flight across the shift"), wrong ? "FAILS" : "ok", wrong);
    }
    return 0;
}
