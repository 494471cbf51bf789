// flow_iter.h -- what flow.hip (plan, launches) and flow_iter.hip (the marching kernel) share.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#ifndef IPX_FI_C_BPERMUTE
#define IPX_FI_C_BPERMUTE 0      // 1: phase C exchanges its pair and quad sums with ds_bpermute (the round-2 form)
#endif
// Ablation builds (tools/fi_ablate.py, timing only, results are garbage): bit 0 = the consumer waves do no work,
// bit 1 = the second expansion is sampled from the pair's first frame (its lines are L1 hits), bit 2 = the
// producers replace the matrix arithmetic by plain sums of what they loaded, bit 3 = one of the four bilinear taps
// is loaded (44 instead of 108 bytes per matrix pixel through L1).
#ifndef IPX_FI_ABLATE
#define IPX_FI_ABLATE 0
#endif

// raw buffer addressing (an SGPR resource, a 32-bit vector offset that alone is range-checked, a scalar offset)
typedef unsigned int ipx_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int ipx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ipx_make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// A 16-byte buffer store with a register as scalar offset.  For this form the compiler keeps no distance between the store and
// a following write to its data registers (the hazard of stores wider than 8 bytes, which it does honour when the scalar offset
// is a constant); on gfx950 the distance is needed all the same: polyexp_kernel, whose next instruction overwrote the data,
// stored garbage in every launch of more than about one workgroup per CU (tests/test_gpu_chunk.py: 270 x 480 and larger).
template <int AUX>
__device__ __forceinline__ void ipx_buffer_store_b128(ipx_u32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voffset, soffset, AUX);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}

// floats per frame of an expansion: 5 per pixel, rounded up so that every frame's float4 plane is 16-byte aligned
__host__ __device__ inline size_t r_stride(size_t P) { return (5 * P + 3) & ~(size_t)3; }

constexpr int FI_M = 7, FI_MC = 128, FI_G = 4, FI_TW = FI_MC - 2 * FI_M, FI_THREADS = 512;
constexpr int FI_RING = 23, FI_MROW = 5 * FI_MC, FI_HALF = FI_MC / 2;
constexpr size_t FI_LDS_BYTES = FI_RING * FI_MROW * sizeof(float) + 5 * FI_G * FI_MC * sizeof(double) + 16;   // + hand-off counter

// Where the flow vectors of an iteration come from / go to.  IN: 0 = flow_in of the same level, 1 = zero
// (first iteration of the coarsest level), 2 = linear resize x2 of the coarser level's flow, evaluated in
// place of a separate upsampling pass (flow of a level = linear resize of the coarser flow, times
// 1/pyr_scale = 2).  REDUCE: the last level-0 iteration of the metric path does not store the field at
// all; it accumulates the f32 magnitudes (ip.py:868) into one f64 partial per workgroup.
struct FiArgs {
    const float *R;
    const float2 *flow_in;      // IN 0: [np][h][w]; IN 2: the coarser field [np][sh][sw]
    float2 *flow_out;
    double *partial;            // REDUCE: [np][strips * nseg]
    int h, w, seg_rows, strips, nseg, np, ppx;
    int sgroup;                         // strips of a pair that are resident together (workgroup order; 1 = one strip of ppx pairs)
    int sh, sw;
    const int *x_ofs, *y_ofs;
    const float *x_a, *y_b;
    unsigned *err;              // device error word of the flow stage (bit 0: a consumer hand-off timed out)
    unsigned spin_limit;        // polls a consumer wave waits for its partners before it gives up
    int prio;                   // probe: raise the waves' issue priority
};

// flow_iter.hip.  in_mode / reduce select the template variant (FiArgs above); `grid` workgroups of FI_THREADS threads
// and FI_LDS_BYTES of dynamic LDS on `stream`.
void ipx_flow_iter_launch(hipStream_t stream, int in_mode, bool reduce, dim3 grid, const FiArgs &a);
// raises the dynamic-LDS limit of the four variants (once per process and device context)
hipError_t ipx_flow_iter_prepare();
