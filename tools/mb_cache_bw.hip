// Cache-level read bandwidth of one MI355X by working-set size: the roof of the cache-resident configurations
// (BASELINE C1 / C2: a few MB of tables; C4: ~77 MB of tables + adjacency) is the L2 / Infinity-Cache rate, not the 8 TB/s of HBM.
//   stream : every lane reads float4s, grid-stride, 8 requests in flight; the buffer is re-read by every launch (resident in whatever
//            cache level holds it)                                              -> the most a kernel can read out of that footprint
//   gather : lane groups of ROWB/16 lanes read uniformly random ROWB-byte rows (256 B = D 64, 512 B = D 128), 8 rows in flight
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mb_cache_bw.hip -o /tmp/mb_cache_bw && /tmp/mb_cache_bw
// Prints one JSON object per footprint; tools/profile_r04.sh files it as profiles/r04_mb_cache_bw.txt.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ x, int64_t n4, float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    float acc = 0.f;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = x[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    for (; i < n4; i += stride) { const float4 v = x[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;          // keep the loads
}

template <int LPR>                               // lanes per row; a row = LPR float4
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ x, const uint32_t* __restrict__ ids, int64_t B, float* __restrict__ out) {
    constexpr int GPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t g0 = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR, TG = (int64_t)gridDim.x * GPB;
    float acc = 0.f;
    for (int64_t b = g0; b + 7 * TG < B; b += 8 * TG) {
        uint32_t r[8]; float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = ids[b + j * TG];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = x[(int64_t)r[j] * LPR + sub];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const int64_t B = 1 << 21;
    float* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double sizes_mb[] = {1, 2, 4, 8, 16, 24, 32, 48, 77, 128, 192, 256, 512, 2048, 16384};
    for (double mb : sizes_mb) {
        const int64_t bytes = (int64_t)(mb * 1e6) / 512 * 512;
        float4* x; CK(hipMalloc(&x, bytes)); CK(hipMemset(x, 0, bytes));
        const int64_t n4 = bytes / 16;
        float ms;
        int grid = 256 * 8;
        for (int w = 0; w < 3; ++w) stream_kernel<<<grid, 256>>>(x, n4, out);
        const int reps = 40;
        CK(hipEventRecord(e0));
        for (int w = 0; w < reps; ++w) stream_kernel<<<grid, 256>>>(x, n4, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double stream = (double)bytes / (ms / reps * 1e-3) / 1e9;
        double gat[2];
        for (int v = 0; v < 2; ++v) {
            const int rowb = v == 0 ? 256 : 512;
            const int64_t rows = bytes / rowb;
            std::vector<uint32_t> h(B);
            uint64_t s = 88172645463325252ull;
            for (int64_t i = 0; i < B; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % (uint64_t)rows); }
            uint32_t* ids; CK(hipMalloc(&ids, B * 4)); CK(hipMemcpy(ids, h.data(), B * 4, hipMemcpyHostToDevice));
            auto launch = [&]() { if (v == 0) gather_kernel<16><<<grid, 256>>>(x, ids, B, out); else gather_kernel<32><<<grid, 256>>>(x, ids, B, out); };
            for (int w = 0; w < 3; ++w) launch();
            CK(hipEventRecord(e0));
            for (int w = 0; w < reps; ++w) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            gat[v] = (double)(B / (8 * (int64_t)grid * (256 / (rowb / 16))) * (8 * (int64_t)grid * (256 / (rowb / 16)))) * rowb / (ms / reps * 1e-3) / 1e9;
            CK(hipFree(ids));
        }
        printf("{\"footprint_MB\": %.1f, \"stream_read_GBps\": %.0f, \"random_256B_row_gather_GBps\": %.0f, \"random_512B_row_gather_GBps\": %.0f}\n",
               bytes / 1e6, stream, gat[0], gat[1]);
        fflush(stdout);
        CK(hipFree(x));
    }
    return 0;
}
