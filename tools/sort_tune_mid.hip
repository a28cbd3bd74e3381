// rocPRIM configurations for the id sort of a 65,536-triple step: 196,608 (key, index) pairs, 27 significant key bits.  The library default
// at this size is its merge sort: one block sort of 1,024-item runs + 8 merge passes (9 launches, ~58 us in the step's trace).
// Build and run on an MI355X:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sort_tune_mid.hip -o /tmp/sort_tune_mid && /tmp/sort_tune_mid
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
    for (int i = 0; i < 200; ++i) CK((rocprim::radix_sort_pairs<Config>(t, tmp, kin, kout, vin, vout, n, 0u, bits)));
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> k(n), v(n); CK(hipMemcpy(k.data(), kout, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(v.data(), vout, n * 4, hipMemcpyDeviceToHost));
    bool ok = true; for (size_t i = 1; i < n; ++i) if (k[i - 1] > k[i] || (k[i - 1] == k[i] && v[i - 1] > v[i])) { ok = false; break; }   // sorted AND stable
    printf("%-44s %.1f us  sorted+stable=%d tmp=%zu\n", name, ms / 200 * 1e3, (int)ok, tmp);
    hipFree(t); return 0;
}
int main(int argc, char** argv) {
    const size_t B = argc > 1 ? (size_t)atol(argv[1]) : 65536;
    const size_t n = 3 * B; const unsigned bits = 27;
    std::vector<uint32_t> hk(n), hv(n); std::mt19937 g(1);
    for (size_t i = 0; i < n; ++i) { hk[i] = (i < B ? g() % 50000001u : (1u << 26) | (g() % 20000001u)); hv[i] = (uint32_t)i; }
    uint32_t *kin, *kout, *vin, *vout;
    CK(hipMalloc(&kin, n * 4)); CK(hipMalloc(&kout, n * 4)); CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
    CK(hipMemcpy(kin, hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vin, hv.data(), n * 4, hipMemcpyHostToDevice));
    using namespace rocprim;
    printf("n = %zu pairs\n", n);
    run<default_config>("default", kin, kout, vin, vout, n, bits);
#define MS(OE, SB, IPT, PB, MB, MI, MINMP) run<radix_sort_config<default_config, merge_sort_config<OE, SB, IPT, PB, MB, MI, MINMP>, default_config, (size_t)1 << 20>>( \
        "merge oe" #OE " sort" #SB "x" #IPT " mp" #MB "x" #MI " min" #MINMP, kin, kout, vin, vout, n, bits)
    MS(512, 256, 4, 128, 128, 4, 201072);
    MS(512, 512, 4, 128, 128, 4, 201072);
    MS(512, 512, 8, 128, 128, 4, 201072);
    MS(512, 1024, 4, 128, 128, 4, 201072);
    MS(512, 1024, 8, 128, 128, 4, 201072);
    MS(256, 1024, 8, 128, 128, 4, 201072);
    MS(1024, 1024, 8, 128, 128, 4, 201072);
    MS(512, 1024, 8, 128, 128, 4, 1024);          // merge-path passes instead of odd-even
    MS(512, 1024, 8, 128, 256, 8, 1024);
    MS(512, 1024, 8, 128, 512, 8, 1024);
    MS(512, 512, 8, 128, 256, 8, 1024);
    MS(512, 1024, 4, 128, 256, 4, 1024);
#define OS(B_, IPT, R, ALG) run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<B_, IPT>, kernel_config<B_, IPT>, R, block_radix_rank_algorithm::ALG>, (size_t)1 << 12>>( \
        "onesweep " #B_ "x" #IPT " r" #R " " #ALG, kin, kout, vin, vout, n, bits)
    OS(1024, 8, 9, match); OS(512, 8, 9, match); OS(256, 8, 9, match); OS(256, 4, 9, match); OS(512, 4, 9, match); OS(1024, 4, 9, match); OS(256, 12, 7, match);
    return 0;
}
