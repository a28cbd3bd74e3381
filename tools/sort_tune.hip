// rocPRIM Onesweep configurations for the id sort of the C5 step: 3,145,728 (key, index) pairs, 27 significant key bits.
// Build and run on an MI355X:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sort_tune.hip -o /tmp/sort_tune && /tmp/sort_tune
// Measured (round 2): library default 161 us; 1,024 threads x 8 items, 9-bit digits, match ranking 117 us (csrc/cdr_step.hip uses it).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class Config>
int run(const char* name, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned bits) {
    size_t tmp = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tmp, kin, kout, vin, vout, n, 0u, bits)));
    void* t; CK(hipMalloc(&t, tmp));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) CK((rocprim::radix_sort_pairs<Config>(t, tmp, kin, kout, vin, vout, n, 0u, bits)));
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) CK((rocprim::radix_sort_pairs<Config>(t, tmp, kin, kout, vin, vout, n, 0u, bits)));
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> k(n); CK(hipMemcpy(k.data(), kout, n * 4, hipMemcpyDeviceToHost));
    bool ok = true; for (size_t i = 1; i < n; ++i) if (k[i - 1] > k[i]) { ok = false; break; }
    printf("%-40s %.1f us  sorted=%d tmp=%zu\n", name, ms / 50 * 1e3, (int)ok, tmp);
    hipFree(t); return 0;
}
int main() {
    const size_t n = 3 * 1048576; const unsigned bits = 27;
    std::vector<uint32_t> hk(n), hv(n); std::mt19937 g(1);
    for (size_t i = 0; i < n; ++i) { hk[i] = (i < 1048576 ? g() % 50000001u : (1u << 26) | (g() % 20000001u)); hv[i] = (uint32_t)i; }
    uint32_t *kin, *kout, *vin, *vout;
    CK(hipMalloc(&kin, n * 4)); CK(hipMalloc(&kout, n * 4)); CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
    CK(hipMemcpy(kin, hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vin, hv.data(), n * 4, hipMemcpyHostToDevice));
    using namespace rocprim;
    run<default_config>("default", kin, kout, vin, vout, n, bits);
#define OS(B, IPT, R, ALG) run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<B, IPT>, kernel_config<B, IPT>, R, block_radix_rank_algorithm::ALG>>>(#B "x" #IPT " r" #R " " #ALG, kin, kout, vin, vout, n, bits)
    OS(256, 12, 8, match); OS(256, 12, 9, match); OS(512, 12, 9, match); OS(512, 12, 8, match); OS(768, 12, 9, match);
    OS(1024, 8, 9, match); OS(1024, 6, 9, match); OS(1024, 7, 9, match); OS(1024, 9, 9, match); OS(1024, 10, 9, match); OS(1024, 12, 9, match); OS(1024, 16, 9, match);
    OS(1024, 8, 10, match); OS(1024, 8, 7, match);
    return 0;
}
