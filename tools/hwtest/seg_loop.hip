// Stand-alone reproduction of ccl.hip's per-word segment loop (out must equal in), with variants.
// Every index stays inside its array even if a segment is mis-computed: nothing here can fault.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ void first_segment(uint64_t rem, int &s, int &len, uint64_t &mask)
{
    s = __builtin_ctzll(rem);
    const uint64_t inv = ~(rem >> s);
    len = inv ? __builtin_ctzll(inv) : 64;
    if (len > 64 - s) len = 64 - s;
    mask = (len == 64 ? ~0ull : ((1ull << len) - 1)) << s;
}

// V = 0: no memory operation inside the loop; 1: dependent loads (as select_untainted_kernel);
//     2: as 1 with the loads drained before the next vector ALU instruction; 3: stores (as ccl_init_kernel)
template <int V, bool DBG, bool PAD = false>
__global__ void seg_kernel(const uint64_t *bits, size_t nw, int words, const int *par, int *aux0, uint64_t *out, unsigned long long *dbg)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (PAD) asm volatile("; keep one more register allocated" ::: "v16");
    if (g >= nw) return;
    uint64_t rem = bits[g];
    uint64_t r = 0;
    if (rem) {
        const size_t plane = g / (size_t)words;
        const int w = (int)(g - plane * words);
        const int *pp = par + plane * (size_t)words * 64;
        int *a0 = aux0 + plane * (size_t)words * 64;
        const uint64_t orig = rem;
        int iter = 0;
        while (rem) {
            int s, len; uint64_t m;
            const uint64_t rem_in = rem;
            first_segment(rem, s, len, m);
            if (DBG && (m & ~orig) && dbg) {                    // a mask with bits the word does not have
                const unsigned long long slot = atomicAdd(dbg, 1ull);
                if (slot < 16) {
                    unsigned long long *d = dbg + 1 + slot * 8;
                    d[0] = g; d[1] = orig; d[2] = rem_in; d[3] = m; d[4] = (unsigned long long)s; d[5] = (unsigned long long)len; d[6] = iter; d[7] = threadIdx.x & 63;
                }
            }
            ++iter;
            rem &= ~m;
            const int id = (w << 6) | s;
            if (V == 0) {
                r |= m;
            } else if (V == 1) {
                if (!a0[pp[id]]) r |= m;
            } else if (V == 2) {
                const int root = pp[id];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int t = a0[root];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!t) r |= m;
            } else {
                a0[id] = 0;
                r |= m;
            }
        }
    }
    out[g] = r;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
    const int count = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int words = argc > 3 ? atoi(argv[3]) : 4800;
    const size_t nw = (size_t)words * count, nid = nw * 64;
    std::vector<uint64_t> h(nw, 0), o(nw);
    srand(1);
    for (size_t i = 0; i < nw; ++i)
        if (rand() % 40 == 0) { uint64_t v = 0; const int k = 1 + rand() % 3; for (int j = 0; j < k; ++j) v |= (uint64_t)(1 + rand() % 3) << (rand() % 62); h[i] = v; }
    uint64_t *d_in, *d_out; int *par, *aux0; unsigned long long *dbg;
    CK(hipMalloc(&dbg, 8 * 129)); CK(hipMemset(dbg, 0, 8 * 129));
    CK(hipMalloc(&d_in, nw * 8)); CK(hipMalloc(&d_out, nw * 8));
    CK(hipMalloc(&par, nid * 4)); CK(hipMalloc(&aux0, nid * 4));
    CK(hipMemcpy(d_in, h.data(), nw * 8, hipMemcpyHostToDevice));
    CK(hipMemset(par, 0, nid * 4)); CK(hipMemset(aux0, 0, nid * 4));     // every pp[id] = 0, every a0[..] = 0: all in range
    const dim3 grid((unsigned)((nw + 255) / 256)), block(256);
    for (int v = 0; v < 6; ++v) {
        long bad_total = 0; int shown = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(d_out, 0xee, nw * 8));
            if (v == 0) hipLaunchKernelGGL((seg_kernel<0, false>), grid, block, 0, 0, d_in, nw, words, par, aux0, d_out, dbg);
            if (v == 1) hipLaunchKernelGGL((seg_kernel<1, false>), grid, block, 0, 0, d_in, nw, words, par, aux0, d_out, dbg);
            if (v == 2) hipLaunchKernelGGL((seg_kernel<2, false>), grid, block, 0, 0, d_in, nw, words, par, aux0, d_out, dbg);
            if (v == 3) hipLaunchKernelGGL((seg_kernel<3, false>), grid, block, 0, 0, d_in, nw, words, par, aux0, d_out, dbg);
            if (v == 5) hipLaunchKernelGGL((seg_kernel<1, false, true>), grid, block, 0, 0, d_in, nw, words, par, aux0, d_out, dbg);
            if (v == 4) hipLaunchKernelGGL((seg_kernel<1, true>), grid, block, 0, 0, d_in, nw, words, par, aux0, d_out, dbg);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(o.data(), d_out, nw * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nw; ++i)
                if (o[i] != h[i]) {
                    if (shown < 3) { printf("  variant %d rep %d word %zu (lane %zu): got %016llx want %016llx\n", v, r, i, i & 63, (unsigned long long)o[i], (unsigned long long)h[i]); ++shown; }
                    ++bad_total;
                }
        }
        printf("variant %d: count %d reps %d: total wrong words %ld\n", v, count, reps, bad_total);
        unsigned long long hd[129];
        CK(hipMemcpy(hd, dbg, sizeof(hd), hipMemcpyDeviceToHost));
        for (unsigned long long i = 0; i < hd[0] && i < 6; ++i) {
            const unsigned long long *d = hd + 1 + i * 8;
            printf("   glitch g=%llu (g&63=%llu lane=%llu w&63=%llu) orig=%016llx rem_in=%016llx m=%016llx s=%llu len=%llu iter=%llu\n", d[0], d[0] & 63, d[7],
                   (d[0] % words) & 63, d[1], d[2], d[3], d[4], d[5], d[6]);
        }
        CK(hipMemset(dbg, 0, 8 * 129));
    }
    return 0;
}
