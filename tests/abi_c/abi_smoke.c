/* A plain C consumer of libcdrhip.so: no Python, no torch -- device buffers from the HIP runtime, entry points from
 * include/cdr_hip.h.  Runs the pairwise loss (cdr_bpr_fwd), the all-items scoring (cdr_fullsort_scores_f32) and one fused
 * row-wise SGD step (cdr_bpr_fwd_grad -> cdr_sort_ids_two_tables -> cdr_rowwise_apply x2; then the same step in the dimension
 * layout's call sequence, cdr_ids_pack32 ... cdr_bpr_grad_from_diff) on small tables and checks every
 * result against the same arithmetic done here in double precision.  Exit code 0 = all within 1e-5 relative.
 * Build: hipcc -x c tests/abi_c/abi_smoke.c -Iinclude -Lrecbole-cdr_amd/lib -lcdrhip -o tests/abi_c/abi_smoke           */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "cdr_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_CDR(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, cdr_last_error()); return 3; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static double urand(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0; }
static int close_enough(double got, double want, double scale, const char* what) {
    const double tol = 1e-5 * (fabs(want) > scale ? fabs(want) : scale);
    if (fabs(got - want) > tol) { fprintf(stderr, "MISMATCH %s: got %.9g want %.9g\n", what, got, want); return 0; }
    return 1;
}

int main(void) {
    enum { NU = 500, NI = 300, D = 64, B = 2000 };
    const float gamma = 1e-10f, reg = 0.01f, lr = 0.05f;
    float* U = (float*)malloc(sizeof(float) * NU * D); float* I = (float*)malloc(sizeof(float) * NI * D);
    int64_t *u = (int64_t*)malloc(8 * B), *p = (int64_t*)malloc(8 * B), *n = (int64_t*)malloc(8 * B);
    for (int i = 0; i < NU * D; ++i) U[i] = (float)(urand() - 0.5) * 0.4f;
    for (int i = 0; i < NI * D; ++i) I[i] = (float)(urand() - 0.5) * 0.4f;
    for (int b = 0; b < B; ++b) { u[b] = (int64_t)(urand() * NU); p[b] = (int64_t)(urand() * NI); n[b] = (int64_t)(urand() * NI); }
    if (cdr_abi_version() != CDR_ABI_VERSION) { fprintf(stderr, "library ABI %d, header ABI %d\n", cdr_abi_version(), CDR_ABI_VERSION); return 4; }

    /* ---- host reference in double ---------------------------------------------------------------------------- */
    double loss = 0, su = 0, sp = 0;
    double* g = (double*)malloc(sizeof(double) * B);
    for (int b = 0; b < B; ++b) {
        double dp = 0, dn = 0;
        for (int d = 0; d < D; ++d) {
            const double uu = U[u[b] * D + d], pp = I[p[b] * D + d], nn = I[n[b] * D + d];
            dp += uu * pp; dn += uu * nn; su += uu * uu; sp += pp * pp;
        }
        const double s = 1.0 / (1.0 + exp(-(dp - dn)));
        loss += -log((double)gamma + s);
        g[b] = -(1.0 / B) * (s * (1.0 - s)) / ((double)gamma + s);
    }
    const double main_loss = loss / B, nu_ = sqrt(su), ni_ = sqrt(sp), total = main_loss + reg * (nu_ + ni_) / B;
    const double cu = reg / (B * nu_), ci = reg / (B * ni_);
    double* Uref = (double*)malloc(sizeof(double) * NU * D); double* Iref = (double*)malloc(sizeof(double) * NI * D);
    for (int i = 0; i < NU * D; ++i) Uref[i] = U[i];
    for (int i = 0; i < NI * D; ++i) Iref[i] = I[i];
    for (int b = 0; b < B; ++b)                                   /* SGD on the summed gradients at the pre-step weights */
        for (int d = 0; d < D; ++d) {
            const double uu = U[u[b] * D + d], pp = I[p[b] * D + d], nn = I[n[b] * D + d];
            Uref[u[b] * D + d] -= lr * (g[b] * (pp - nn) + cu * uu);
            Iref[p[b] * D + d] -= lr * (g[b] * uu + ci * pp);
            Iref[n[b] * D + d] -= lr * (-g[b] * uu);
        }

    /* ---- device side through the C ABI ------------------------------------------------------------------------ */
    float *dU, *dI, *dOut, *dGU, *dGP, *dScores; int64_t *du, *dp_, *dn_; uint32_t *dKeys, *dPerm; void* dWs;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipMalloc((void**)&dU, sizeof(float) * NU * D)); CHECK_HIP(hipMalloc((void**)&dI, sizeof(float) * NI * D));
    CHECK_HIP(hipMalloc((void**)&dOut, sizeof(float) * 16)); CHECK_HIP(hipMalloc((void**)&dGU, sizeof(float) * B * D));
    CHECK_HIP(hipMalloc((void**)&dGP, sizeof(float) * B * D)); CHECK_HIP(hipMalloc((void**)&dScores, sizeof(float) * 3 * NI));
    CHECK_HIP(hipMalloc((void**)&du, 8 * B)); CHECK_HIP(hipMalloc((void**)&dp_, 8 * B)); CHECK_HIP(hipMalloc((void**)&dn_, 8 * B));
    CHECK_HIP(hipMalloc((void**)&dKeys, 4 * 3 * B)); CHECK_HIP(hipMalloc((void**)&dPerm, 4 * 3 * B));
    CHECK_HIP(hipMemcpy(dU, U, sizeof(float) * NU * D, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dI, I, sizeof(float) * NI * D, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(du, u, 8 * B, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dp_, p, 8 * B, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dn_, n, 8 * B, hipMemcpyHostToDevice)); CHECK_HIP(hipMemset(dOut, 0, sizeof(float) * 16));
    cdr_ctx* ctx = NULL;
    fprintf(stderr, "abi_smoke: stage ctx\n"); fflush(stderr);
    CHECK_CDR(cdr_ctx_create(0, &ctx));
    float out[16];
    int ok = 1;

    CHECK_CDR(cdr_bpr_fwd(ctx, NULL, dU, dI, D, du, dp_, dn_, B, gamma, reg, dOut, NULL));
    CHECK_HIP(hipDeviceSynchronize()); CHECK_HIP(hipMemcpy(out, dOut, sizeof(float) * 4, hipMemcpyDeviceToHost));
    ok &= close_enough(out[0], total, 0, "bpr_fwd total loss") & close_enough(out[1], main_loss, 0, "bpr_fwd main loss") &
          close_enough(out[2], nu_, 0, "EmbLoss user norm") & close_enough(out[3], ni_, 0, "EmbLoss item norm");

    CHECK_CDR(cdr_fullsort_scores_f32(NULL, dU, 3, D, dI, NI, NULL, 0, dScores));
    float* sc = (float*)malloc(sizeof(float) * 3 * NI);
    CHECK_HIP(hipDeviceSynchronize()); CHECK_HIP(hipMemcpy(sc, dScores, sizeof(float) * 3 * NI, hipMemcpyDeviceToHost));
    for (int q = 0; q < 3; ++q)
        for (int j = 0; j < NI; j += 37) {
            double want = 0;
            for (int d = 0; d < D; ++d) want += (double)U[q * D + d] * I[j * D + d];
            ok &= close_enough(sc[q * NI + j], want, 0.05, "full-sort score");
        }

    size_t ws_bytes = 0;
    fprintf(stderr, "abi_smoke: stage two-pass step\n"); fflush(stderr);
    CHECK_CDR(cdr_sort_workspace_bytes(3 * B, 2048, &ws_bytes));            /* keys < 2 * 2^ceil(log2(max rows)) = 1024 */
    CHECK_HIP(hipMalloc(&dWs, ws_bytes));
    uint32_t key_base = 0;
    CHECK_CDR(cdr_bpr_fwd_grad(ctx, NULL, dU, dI, D, du, dp_, dn_, B, 0, gamma, reg, dOut, dGU, dGP, 0));
    CHECK_CDR(cdr_sort_ids_two_tables(ctx, NULL, du, B, NU, dp_, B, dn_, B, NI, dKeys, dPerm, &key_base, dWs, ws_bytes));
    CHECK_CDR(cdr_rowwise_apply(ctx, NULL, CDR_OPT_SGD, dU, NULL, NULL, D, dKeys, dPerm, B, dGU, B, B, dOut + 4, lr, 0.9f, 0.999f, 1e-8f,
                                0.f, 1, NULL, 0));
    CHECK_CDR(cdr_rowwise_apply(ctx, NULL, CDR_OPT_SGD, dI, NULL, NULL, D, dKeys + B, dPerm + B, 2 * B, dGP, B, B, dOut + 5, lr, 0.9f,
                                0.999f, 1e-8f, 0.f, 1, NULL, key_base));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(out, dOut, sizeof(float) * 6, hipMemcpyDeviceToHost));
    ok &= close_enough(out[0], total, 0, "fused step loss");
    float* U2 = (float*)malloc(sizeof(float) * NU * D); float* I2 = (float*)malloc(sizeof(float) * NI * D);
    CHECK_HIP(hipMemcpy(U2, dU, sizeof(float) * NU * D, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(I2, dI, sizeof(float) * NI * D, hipMemcpyDeviceToHost));
    for (int i = 0; i < NU * D; ++i) ok &= close_enough(U2[i], Uref[i], 0.2, "user table after the step");
    for (int i = 0; i < NI * D; ++i) ok &= close_enough(I2[i], Iref[i], 0.2, "item table after the step");
    /* ---- the same step in the dimension layout's call sequence, one rank holding every column: ids narrowed
     * and widened again around where the all-gather would be, partial scores, where the all-reduce would be, sort, gradients
     * from the scores, the two applies -- on fresh copies of the tables, against the same host reference ------------------- */
    float *dU2, *dI2, *dDiff; int32_t *dIds32; int64_t* dIds64; int* dBad;
    CHECK_HIP(hipMalloc((void**)&dU2, sizeof(float) * NU * D)); CHECK_HIP(hipMalloc((void**)&dI2, sizeof(float) * NI * D));
    CHECK_HIP(hipMemcpy(dU2, U, sizeof(float) * NU * D, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dI2, I, sizeof(float) * NI * D, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc((void**)&dDiff, sizeof(float) * (B + 2))); CHECK_HIP(hipMalloc((void**)&dIds32, 4 * 3 * B));
    CHECK_HIP(hipMalloc((void**)&dIds64, 8 * 3 * B)); CHECK_HIP(hipMalloc((void**)&dBad, 4)); CHECK_HIP(hipMemset(dBad, 0, 4));
    fprintf(stderr, "abi_smoke: stage dimension layout\n"); fflush(stderr);
    CHECK_CDR(cdr_ids_pack32(NULL, du, dp_, dn_, NULL, B, dIds32, dBad));
    /* ncclAllGather(dIds32 -> [G][3][B]) would go here; G = 1 */
    CHECK_CDR(cdr_ids_unpack32(NULL, dIds32, 1, B, dIds64, NULL));
    const int64_t *gu = dIds64, *gp = dIds64 + B, *gn = dIds64 + 2 * B;
    CHECK_CDR(cdr_bpr_partial_diff(ctx, NULL, dU2, dI2, D, gu, gp, gn, B, dDiff));
    /* ncclAllReduce(dDiff, B + 2, sum) would go here */
    CHECK_CDR(cdr_sort_ids_two_tables(ctx, NULL, gu, B, NU, gp, B, gn, B, NI, dKeys, dPerm, &key_base, dWs, ws_bytes));
    CHECK_CDR(cdr_bpr_grad_from_diff(ctx, NULL, dU2, dI2, D, gu, gp, gn, B, gamma, reg, dDiff, dOut, dGU, dGP));
    CHECK_CDR(cdr_rowwise_apply(ctx, NULL, CDR_OPT_SGD, dU2, NULL, NULL, D, dKeys, dPerm, B, dGU, B, B, dOut + 4, lr, 0.9f, 0.999f, 1e-8f,
                                0.f, 1, NULL, 0));
    CHECK_CDR(cdr_rowwise_apply(ctx, NULL, CDR_OPT_SGD, dI2, NULL, NULL, D, dKeys + B, dPerm + B, 2 * B, dGP, B, B, dOut + 5, lr, 0.9f,
                                0.999f, 1e-8f, 0.f, 1, NULL, key_base));
    CHECK_HIP(hipDeviceSynchronize());
    int bad = 1;
    CHECK_HIP(hipMemcpy(&bad, dBad, 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out, dOut, sizeof(float) * 6, hipMemcpyDeviceToHost));
    ok &= (bad == 0) & close_enough(out[0], total, 0, "dimension-layout step loss");
    CHECK_HIP(hipMemcpy(U2, dU2, sizeof(float) * NU * D, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(I2, dI2, sizeof(float) * NI * D, hipMemcpyDeviceToHost));
    for (int i = 0; i < NU * D; ++i) ok &= close_enough(U2[i], Uref[i], 0.2, "user table after the dimension-layout step");
    for (int i = 0; i < NI * D; ++i) ok &= close_enough(I2[i], Iref[i], 0.2, "item table after the dimension-layout step");
    /* ---- the same step as ONE call (cdr_bpr_step_fused: rows that occur once are updated by the forward kernel), on fresh copies -- */
    {
        float *dU3, *dI3; uint8_t* dFlags; uint32_t* dHeads; int64_t words = 0;
        CHECK_HIP(hipMalloc((void**)&dU3, sizeof(float) * NU * D)); CHECK_HIP(hipMalloc((void**)&dI3, sizeof(float) * NI * D));
        CHECK_HIP(hipMemcpy(dU3, U, sizeof(float) * NU * D, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dI3, I, sizeof(float) * NI * D, hipMemcpyHostToDevice));
        fprintf(stderr, "abi_smoke: stage one-call fused step\n"); fflush(stderr);
        CHECK_CDR(cdr_bpr_step_fused_heads_words(B, &words));
        CHECK_HIP(hipMalloc((void**)&dFlags, 4 * B)); CHECK_HIP(hipMalloc((void**)&dHeads, 4 * (size_t)words));
        CHECK_CDR(cdr_bpr_step_fused(ctx, NULL, CDR_OPT_SGD, dU3, NULL, NULL, NU, dI3, NULL, NULL, NI, D, du, dp_, dn_, B, gamma, reg, lr, 0.9f,
                                     0.999f, 1e-8f, 0.f, 1, 1, dOut, dGU, dGP, dKeys, dPerm, dFlags, dHeads, dWs, ws_bytes));
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipMemcpy(out, dOut, sizeof(float) * 6, hipMemcpyDeviceToHost));
        ok &= close_enough(out[0], total, 0, "one-call fused step loss");
        CHECK_HIP(hipMemcpy(U2, dU3, sizeof(float) * NU * D, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(I2, dI3, sizeof(float) * NI * D, hipMemcpyDeviceToHost));
        for (int i = 0; i < NU * D; ++i) ok &= close_enough(U2[i], Uref[i], 0.2, "user table after the one-call fused step");
        for (int i = 0; i < NI * D; ++i) ok &= close_enough(I2[i], Iref[i], 0.2, "item table after the one-call fused step");
    }
    /* ---- family (10): the exchanges of the multi-GPU path from a host without torch -- a one-rank communicator (RCCL from the loader path) ---- */
    {
        unsigned char id[CDR_COMM_ID_BYTES];
        fprintf(stderr, "abi_smoke: stage comm family (RCCL)\n"); fflush(stderr);
        /* CDR_ABI_SMOKE_SKIP_COMM=1: leave the one-rank communicator out (its bootstrap has been seen to stand still on a fresh box) */
        const char* skip = getenv("CDR_ABI_SMOKE_SKIP_COMM");
        int rc = (skip && skip[0] == '1') ? CDR_ENODEV : cdr_comm_unique_id(id);
        if (rc == CDR_ENODEV) printf("abi_smoke: librccl not loadable here, comm family skipped (%s)\n", cdr_last_error());
        else {
            cdr_comm* comm = NULL; int rank = -1, world = -1;
            ok &= rc == 0;
            fprintf(stderr, "abi_smoke: stage comm init\n"); fflush(stderr);
            CHECK_CDR(cdr_comm_init(&comm, 0, 1, id));
            fprintf(stderr, "abi_smoke: stage comm exchanges\n"); fflush(stderr);
            CHECK_CDR(cdr_comm_info(comm, &rank, &world));
            ok &= rank == 0 && world == 1;
            const int64_t nrows[1] = {B};
            CHECK_CDR(cdr_a2a_rows(comm, NULL, dGU, nrows, dGP, nrows, D));           /* rows to "their owner" and back: the identity on one rank */
            CHECK_CDR(cdr_allreduce_sum_f32(comm, NULL, dOut, 6));
            CHECK_HIP(hipDeviceSynchronize());
            float after[6];
            CHECK_HIP(hipMemcpy(after, dOut, sizeof(after), hipMemcpyDeviceToHost));
            ok &= close_enough(after[0], out[0], 0, "loss through a one-rank all-reduce");
            CHECK_CDR(cdr_comm_destroy(comm));
        }
    }
    CHECK_CDR(cdr_ctx_destroy(ctx));
    printf(ok ? "abi_smoke: OK (loss %.7f)\n" : "abi_smoke: FAILED (loss %.7f)\n", out[0]);
    return ok ? 0 : 1;
}
