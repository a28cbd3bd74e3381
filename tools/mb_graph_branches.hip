// Do the parallel branches of a captured two-stream hipGraph run side by side on this ROCm?  Two independent kernels, each a quarter of the
// chip wide and ~100 us long: (1) back to back on one stream, (2) eagerly on two streams, (3) captured as fork / join and replayed.
// hipcc --offload-arch=gfx950 -O3 tools/mb_graph_branches.hip -o /tmp/mb_graph_branches && /tmp/mb_graph_branches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(float* p, int iters) {
    float a = p[threadIdx.x], b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main() {
    float *pa, *pb;
    CK(hipMalloc(&pa, 64 * 256 * 4)); CK(hipMalloc(&pb, 64 * 256 * 4));
    CK(hipMemset(pa, 0, 64 * 256 * 4)); CK(hipMemset(pb, 0, 64 * 256 * 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const int iters = 60000, reps = 50, chain = 4;        // each branch: `chain` kernels in a row
    auto branch = [&](float* p, hipStream_t s) { for (int c = 0; c < chain; ++c) spin<<<dim3(64), dim3(256), 0, s>>>(p, iters); };
    auto timeit = [&](const char* what, auto fn) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-62s %8.1f us per repetition\n", what, ms / reps * 1e3);
    };
    timeit("one branch alone (4 kernels, 64 workgroups each)", [&] { branch(pa, s1); });
    timeit("two branches back to back on one stream", [&] { branch(pa, s1); branch(pb, s1); });
    timeit("two branches on two streams, eager (fork / join events)", [&] {
        CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
        branch(pa, s1); branch(pb, s2);
        CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
    });
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
    branch(pa, s1); branch(pb, s2);
    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    timeit("the same captured as ONE hipGraph (two parallel branches)", [&] { CK(hipGraphLaunch(ge, s1)); });
    hipGraph_t g1; hipGraphExec_t ge1;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    branch(pa, s1); branch(pb, s1);
    CK(hipStreamEndCapture(s1, &g1));
    CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
    timeit("captured as one hipGraph, single chain of 8 kernels", [&] { CK(hipGraphLaunch(ge1, s1)); });
    return 0;
}
