// Random-address atomics on gfx950: throughput of {returning CAS + add} (a hash insert) and of plain non-returning adds over tables of 2^k 16-byte slots.
// hipcc --offload-arch=gfx950 -O3 tools/r06/mb_atomics.hip -o /tmp/mb_atomics && /tmp/mb_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
struct slot { uint32_t key, cnt, off, fill; };
__global__ void insert(const uint32_t* keys, int64_t n, slot* T, uint32_t mask, uint32_t* slot_of) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t k = keys[e] + 1u;
        uint32_t h = mix(k) & mask;
        for (;;) {
            const uint32_t old = atomicCAS(&T[h].key, 0u, k);
            if (old == 0u || old == k) { atomicAdd(&T[h].cnt, 1u); slot_of[e] = h; break; }
            h = (h + 1) & mask;
        }
    }
}
__global__ void addonly(const uint32_t* keys, int64_t n, uint32_t* C, uint32_t mask) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&C[mix(keys[e]) & mask], 1u);
}
__global__ void lookup(const uint32_t* slot_of, int64_t n, const slot* T, uint8_t* flags) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        flags[e] = T[slot_of[e]].cnt == 1u;
}
__global__ void fill(uint32_t* k, int64_t n, uint32_t range, uint32_t seed) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) k[e] = mix((uint32_t)e * 2654435761u + seed) % range;
}
int main() {
    for (int64_t n : {196608ll, 786432ll, 3145728ll}) {
        uint32_t S = 1; while (S < 2 * n) S <<= 1;
        uint32_t *keys, *slot_of, *C; slot* T; uint8_t* flags;
        hipMalloc(&keys, n * 4); hipMalloc(&slot_of, n * 4); hipMalloc(&T, (size_t)S * 16); hipMalloc(&C, (size_t)S * 4); hipMalloc(&flags, n);
        fill<<<1024, 256>>>(keys, n, 60000000u, 7u);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms[4] = {0, 0, 0, 0};
        const int R = 20;
        for (int r = 0; r < R + 2; ++r) {
            float t;
            hipEventRecord(a); hipMemsetAsync(T, 0, (size_t)S * 16); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&t, a, b); if (r >= 2) ms[0] += t;
            hipEventRecord(a); insert<<<(n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256, 256>>>(keys, n, T, S - 1, slot_of); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&t, a, b); if (r >= 2) ms[1] += t;
            hipEventRecord(a); lookup<<<(n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256, 256>>>(slot_of, n, T, flags); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&t, a, b); if (r >= 2) ms[2] += t;
            hipMemsetAsync(C, 0, (size_t)S * 4);
            hipEventRecord(a); addonly<<<(n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256, 256>>>(keys, n, C, S - 1); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&t, a, b); if (r >= 2) ms[3] += t;
        }
        printf("n=%lld slots=%u (%.0f MB): memset %.1f us, insert(CAS+add) %.1f us, lookup %.1f us, add-only(4B table) %.1f us\n", (long long)n, S, S * 16.0 / 1e6,
               1e3 * ms[0] / R, 1e3 * ms[1] / R, 1e3 * ms[2] / R, 1e3 * ms[3] / R);
        hipFree(keys); hipFree(slot_of); hipFree(T); hipFree(C); hipFree(flags);
    }
    return 0;
}
