// How much VALU work does a wave get done while another wave of the SAME SIMD issues back-to-back dependent fp32 MFMAs?
// (the OVERLAP kernel of csrc/cdr_mapstep.hip runs one MFMA wave and one row wave per SIMD: this is its hardware budget)
// Block = 512 threads on one CU: waves 0-3 -> one per SIMD (MFMA role), waves 4-7 -> one per SIMD (VALU role).
// Build + run on an MI355X: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NOPS>
__global__ __launch_bounds__(512, 1) void k(int mode, int prio, int iters, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    float acc = threadIdx.x * 1e-3f;
    f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
    __syncthreads();
    const long long t0 = wall_clock64();
    if (mf) {
        if (mode & 1) {
            float a = acc, b = acc + 1.f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
                    if (NOPS >= 1) asm volatile("s_nop 15");
                    if (NOPS >= 2) asm volatile("s_nop 15");
                    if (NOPS >= 3) asm volatile("s_nop 15");
                    if (NOPS >= 4) asm volatile("s_nop 7");
                }
            }
        }
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        if (mode & 2) {
            float x = acc, y = acc * 0.5f, z = 1.0001f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 64; ++j) { x = x * z + y; y = y * z + x; }      // 128 dependent-pair v_fma per iteration
            }
            acc = x + y;
        }
    }
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = acc; for (int r = 0; r < 16; ++r) s += c[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 512 * 4 * 256); hipMalloc(&cyc, 8 * 8 * 256);
    long long h[8];
    const int iters = 2000;
    auto run = [&](auto kern, const char* name) {
        for (int mode = 1; mode < 4; ++mode) {
            kern<<<1, 512>>>(mode, 1, iters, out, cyc); hipDeviceSynchronize();
            kern<<<1, 512>>>(mode, 1, iters, out, cyc); hipDeviceSynchronize();
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-22s %s: MFMA wave %7.1f us (%d x 16 MFMA), VALU wave %7.1f us (%d x 128 dependent fma)\n", name,
                   mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both     ", h[0] / 100.0, iters, h[4] / 100.0, iters);
        }
    };
    run(k<0>, "back-to-back MFMA");
    run(k<1>, "+ s_nop 15 x1");
    run(k<2>, "+ s_nop 15 x2");
    run(k<3>, "+ s_nop 15 x3");
    run(k<4>, "+ s_nop 15 x3 + 7");
    return 0;
}
